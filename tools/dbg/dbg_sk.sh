cd /root/repo
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
for o in -1 0 1; do for d in 0 1; do echo "== order=$o dbg=$d"; MI355TTS_SK_ORDER=$o MI355TTS_GEMM_DBG=$d MI355TTS_SK=1 MI355TTS_SK_STAGES=3 ITERS=50 python tools/gemm_bench.py custom $SH; done; done
