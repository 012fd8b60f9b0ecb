// aa_race.hip — standalone determinism stress of the fused AA+conv kernel (tools only; compiled on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I csrc tools/ubench/aa_race.hip csrc/aa_conv.hip csrc/aa_act.hip csrc/runtime.hip -o aa_race
//   ./aa_race 64 32768 120                                 two workgroups per CU (the product default): round 2 — a few elements
//                                                          differed in 20-35 % of the runs; round 3 (channel-pair AA math, no op_sel
//                                                          broadcast reads): 0 of 119 runs differ
//   MI355TTS_AACONV_LDS_MIN=83968 ./aa_race 64 32768 120   one workgroup per CU (diagnostic policy): 0 runs differ in either round
// identity 1-tap conv => the output IS the activated tile; N runs must be bit-identical.
#include "common.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
using namespace mi;
int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 64, B = 8, T = argc > 2 ? atoi(argv[2]) : 32768, N = argc > 3 ? atoi(argv[3]) : 12;
    const size_t n = (size_t)B * T * C;
    std::vector<uint16_t> hx(n), hw((size_t)C * C, 0);
    uint32_t s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f * 4.f - 2.f; };
    auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); };
    for (auto& v : hx) v = bf(rnd());
    for (int c = 0; c < C; ++c) hw[(size_t)c * C + c] = bf(1.f);
    std::vector<float> al(C), ib(C), bias(C, 0.f);
    for (int c = 0; c < C; ++c) { al[c] = 0.9f + 0.01f * c; ib[c] = 1.0f / (1.0f + 0.02f * c); }
    void *dx, *dw, *dy; float *da, *dib, *db;
    hipMalloc(&dx, n * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dy, n * 2);
    hipMalloc(&da, C * 4); hipMalloc(&dib, C * 4); hipMalloc(&db, C * 4);
    hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(da, al.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(dib, ib.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), C * 4, hipMemcpyHostToDevice);
    hipStream_t st; hipStreamCreate(&st);
    AAConv a;
    a.dtype = MI_BF16; a.x = dx; a.w = dw; a.bias = db; a.snake_alpha = da; a.snake_inv_beta = dib; a.out = dy; a.res = nullptr;
    a.B = B; a.T = T; a.C = C; a.k = 1; a.dil = 1;
    std::vector<uint16_t> ref(n), cur(n);
    int bad_runs = 0, shown = 0; long bad_elems = 0;
    long h_t16[16] = {0}, h_c2[2] = {0}, h_chalf[2] = {0}, h_wave[4] = {0}, h_cq[4] = {0};
    for (int r = 0; r < N; ++r) {
        hipMemsetAsync(dy, 0, n * 2, st);
        launch_aa_conv(a, st);
        hipStreamSynchronize(st);
        hipMemcpy(r == 0 ? ref.data() : cur.data(), dy, n * 2, hipMemcpyDeviceToHost);
        if (r) {
            long d = 0;
            for (size_t i = 0; i < n; ++i)
                if (cur[i] != ref[i]) {
                    ++d;
                    const int c = (int)(i % C); const long bt = (long)(i / C); const int t = (int)(bt % T);
                    ++h_t16[t & 15]; ++h_c2[c & 1]; ++h_chalf[c >= C / 2]; ++h_wave[(t % 128) / 32]; ++h_cq[(c * 4) / C];
                    if (shown < 12) {
                        auto f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; };
                        printf("   run %d b=%ld t=%d (t%%128=%d) c=%d: run0 %.7f this %.7f\n", r, bt / T, t, t % 128, c, f(ref[i]), f(cur[i]));
                        ++shown;
                    }
                }
            bad_runs += d != 0; bad_elems += d;
        }
    }
    printf("  t%%16 histogram:"); for (int i = 0; i < 16; ++i) printf(" %ld", h_t16[i]);
    printf("\n  c odd/even: %ld/%ld  c upper/lower half: %ld/%ld  channel quarter: %ld %ld %ld %ld  row-block(t%%128/32): %ld %ld %ld %ld\n", h_c2[1], h_c2[0], h_chalf[1], h_chalf[0],
           h_cq[0], h_cq[1], h_cq[2], h_cq[3], h_wave[0], h_wave[1], h_wave[2], h_wave[3]);
    printf("C=%d T=%d: %d/%d runs differ from run 0, %ld elements\n", C, T, bad_runs, N - 1, bad_elems);
    return 0;
}
