set -u
O=gpurun_out/r4_b; mkdir -p $O
B="python bench.py --no-secondary --no-cpu-baseline --no-pmc"
for i in 1 2; do
timeout 600 $B --steps 10 --warmup 3 > $O/f32_fold_$i.json 2>$O/err.log
timeout 600 $B --steps 10 --warmup 3 --no-adaln-fold > $O/f32_nofold_$i.json 2>>$O/err.log
done
timeout 600 $B --dtype bf16 --batch 8 --steps 6 --warmup 2 > $O/bf16_fold.json 2>>$O/err.log
timeout 600 $B --dtype bf16 --batch 8 --steps 6 --warmup 2 --no-adaln-fold > $O/bf16_nofold.json 2>>$O/err.log
tail -5 $O/err.log
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "ERR", e); continue
    print(f.split("/")[-1], round(d["ms_per_step"],2), round(d["value"],2), d["config"].get("adaln_fold"), d["config"].get("arithmetic_kind"))
    for k in d["roofline"]["kernels"][:7]: print("    ",k["kernel"][:70],round(k["ms_per_step"],2),k["launches_per_step"],round(k["avg_launch_us"],1))
PY
