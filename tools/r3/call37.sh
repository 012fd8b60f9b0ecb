#!/bin/bash
# per-shape kernel times of the BigVGAN fp16 (8,100,512) forward on one stream (rocprofv3 kernel trace aggregated by grid)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/bvt -- python $ROOT/bench.py --workload bigvgan --steps 4 --warmup 2 --no-cpu-baseline --option bigvgan_streams=1 > /tmp/bvt.log 2>&1
f=$(ls /tmp/bvt/*/*kernel_trace.csv | head -1)
python - "$f" <<'P'
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name']
    if 'dma3' in name: short = 'gemm_dma3'
    elif 'conv_gemm_dma' in name: short = 'gemm_dma2'
    elif 'conv_gemm' in name: short = 'gemm_old'
    elif 'aa_conv' in name: short = 'aa_conv'
    elif 'aa_act' in name: short = 'aa_act'
    else: short = name[:28]
    key = (short, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r.get('LDS_Block_Size', r.get('LDS_Block_Size_v', '')))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(key, [0, 0.0, 1e9, 0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(v[1] for v in agg.values())
nf = 4 + 2 + 1 + 6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k[0]:12s} grid({k[1]},{k[2]},{k[3]}) lds {k[4]:>7s} n={v[0]:5d} avg {v[1]/v[0]:8.1f} us (min {v[2]:.1f} max {v[3]:.1f}) {100*v[1]/tot:5.1f} %")
print(f"kernel time total {tot/1e3:.1f} ms over {nf} forwards")
P
