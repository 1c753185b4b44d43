// Stand-alone harness for contrad_amd/csrc/wino44.h (Winograd F(4x4, 3x3), round 6): correctness against an fp64 direct
// convolution on small shapes (forward with bias + LeakyReLU + addend, data gradient with the activation mask), then the
// time of the kernel on the two gate shapes of the F(2x2, 3x3) prototype (1536 x 16^2 x 128 -> 128, 48 x 128^2 x 128 -> 128)
// and on the other layer shapes of the headline / StyleGAN2_512 steps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wino44_proto.bin tools/micro/wino44_proto.hip && ./wino44_proto.bin [reps] [n|z]
#include "../../contrad_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>
#include <algorithm>
enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
#include "../../contrad_amd/csrc/wino44.h"

static int g_data = 0;      // 0: uniform(-1, 1), 1: normal-ish, 2: zeros

static wino44::Args make_args(int N, int H, int W, int Cin, int Cout) {
  wino44::Args a{};
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldi = Cin; a.ldo = Cout;
  a.TW = std::min(8, W / 4); a.TH = std::min(4, H / 4);
  a.sh_tw = __builtin_ctz(a.TW); a.sh_thw = __builtin_ctz(a.TH * a.TW);
  a.NIMG = 32 / (a.TH * a.TW);
  a.PH = H / (4 * a.TH); a.PW = W / (4 * a.TW);
  a.NP = ((N + a.NIMG - 1) / a.NIMG) * a.PH * a.PW;
  a.NKB = Cout / 64;
  a.BH = 4 * a.TH + 2; a.BW = 4 * a.TW + 2;
  return a;
}

static bool g_filter = true;      // (timing: false = the main kernel alone)
template <int MODE>
static void launch(const wino44::Args& a, const float* wp, float* U, int C, int K, hipStream_t s) {
  const int cin = MODE == MODE_FWD ? C : K, cout = MODE == MODE_FWD ? K : C;
  const int quads = (cin / 4) * cout;
  if (g_filter) hipLaunchKernelGGL(wino44::wino44_filter_kernel<MODE>, dim3((quads + 255) / 256), dim3(256), 0, s, wp, U, C, K, K);
  const int l0 = ((a.NP + 7) / 8) * a.NKB;
  const int grid = 8 * std::min(32, l0);
  if (a.BW == 34) {
    (void)hipFuncSetAttribute((const void*)wino44::wino44_kernel<MODE, 34>, hipFuncAttributeMaxDynamicSharedMemorySize, wino44::LDS_DWORDS * 4);
    hipLaunchKernelGGL((wino44::wino44_kernel<MODE, 34>), dim3(grid), dim3(512), wino44::LDS_DWORDS * 4, s, a);
  } else if (a.BW == 10) {
    (void)hipFuncSetAttribute((const void*)wino44::wino44_kernel<MODE, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, wino44::LDS_DWORDS * 4);
    hipLaunchKernelGGL((wino44::wino44_kernel<MODE, 10>), dim3(grid), dim3(512), wino44::LDS_DWORDS * 4, s, a);
  } else {
    (void)hipFuncSetAttribute((const void*)wino44::wino44_kernel<MODE, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, wino44::LDS_DWORDS * 4);
    hipLaunchKernelGGL((wino44::wino44_kernel<MODE, 18>), dim3(grid), dim3(512), wino44::LDS_DWORDS * 4, s, a);
  }
}

static float frand() {
  static unsigned s = 12345u;
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xFFFF) / 32768.f - 1.f;
}

// mode 0: y = lrelu(conv(x, w) + bias) * gain + addend;   mode 1: dx = conv^T(gy, w) * act'(ref)
static double check(int mode, int N, int H, int W, int C, int K) {
  const int cin = mode == 0 ? C : K, cout = mode == 0 ? K : C;
  std::vector<float> in((size_t)N * H * W * cin), wp((size_t)9 * C * K), bias(K), ref((size_t)N * H * W * cout), out((size_t)N * H * W * cout);
  for (auto& v : in) v = frand();
  for (auto& v : wp) v = frand() / std::sqrt(9.f * cin);
  for (auto& v : bias) v = 0.1f * frand();
  for (auto& v : ref) v = frand();
  float *din, *dwp, *dbias, *dref, *dout, *dU;
  (void)hipMalloc(&din, in.size() * 4); (void)hipMalloc(&dwp, wp.size() * 4); (void)hipMalloc(&dbias, K * 4);
  (void)hipMalloc(&dref, ref.size() * 4); (void)hipMalloc(&dout, out.size() * 4); (void)hipMalloc(&dU, (size_t)36 * C * K * 4);
  (void)hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dwp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dbias, bias.data(), K * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dref, ref.data(), ref.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemset(dout, 0xFF, out.size() * 4);
  wino44::Args a = make_args(N, H, W, cin, cout);
  a.x = din; a.U = dU; a.y = dout; a.bias = mode == 0 ? dbias : nullptr; a.ref = dref; a.slope = 0.2f; a.gain = 1.41421356f;
  if (mode == 0) launch<MODE_FWD>(a, dwp, dU, C, K, 0); else launch<MODE_DGRAD>(a, dwp, dU, C, K, 0);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
  (void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0, worst = 0;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w)
        for (int o = 0; o < cout; ++o) {
          double s = 0;
          for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
              const int hh = mode == 0 ? h + kh - 1 : h - kh + 1, ww = mode == 0 ? w + kw - 1 : w - kw + 1;
              if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
              const float* xi = &in[((size_t)(n * H + hh) * W + ww) * cin];
              for (int c = 0; c < cin; ++c) {
                // packed weight Wp[(kh * 3 + kw) * C + c_in_of_layer][k_out_of_layer]
                const double wv = mode == 0 ? wp[((size_t)(kh * 3 + kw) * C + c) * K + o] : wp[((size_t)(kh * 3 + kw) * C + o) * K + c];
                s += (double)xi[c] * wv;
              }
            }
          const size_t idx = ((size_t)(n * H + h) * W + w) * cout + o;
          double r;
          if (mode == 0) { s += bias[o]; r = s * (s > 0 ? a.gain : a.gain * a.slope) + ref[idx]; }
          else r = s * (ref[idx] > 0 ? a.gain : a.gain * a.slope);
          const double dv = out[idx] - r;
          num += dv * dv; den += r * r; worst = std::max(worst, std::fabs(dv));
        }
  (void)hipFree(din); (void)hipFree(dwp); (void)hipFree(dbias); (void)hipFree(dref); (void)hipFree(dout); (void)hipFree(dU);
  const double rel = std::sqrt(num / den);
  printf("check %s N=%d H=%d W=%d C=%d K=%d: rel-L2 %.3e  max abs %.3e  %s\n", mode == 0 ? "FWD  " : "DGRAD", N, H, W, C, K, rel, worst, rel < 2e-5 ? "ok" : "FAIL");
  return rel;
}

static void timeit(int N, int H, int W, int C, int K, int reps) {
  const size_t nin = (size_t)N * H * W * C, nout = (size_t)N * H * W * K;
  std::vector<float> in(nin), wp((size_t)9 * C * K);
  if (g_data == 2) { std::fill(in.begin(), in.end(), 0.f); std::fill(wp.begin(), wp.end(), 0.f); }
  else { for (auto& v : in) v = frand(); for (auto& v : wp) v = frand() / std::sqrt(9.f * C); }
  float *din, *dwp, *dout, *dU;
  (void)hipMalloc(&din, nin * 4); (void)hipMalloc(&dwp, wp.size() * 4); (void)hipMalloc(&dout, nout * 4); (void)hipMalloc(&dU, (size_t)36 * C * K * 4);
  (void)hipMemcpy(din, in.data(), nin * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dwp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
  wino44::Args a = make_args(N, H, W, C, K);
  a.x = din; a.U = dU; a.y = dout; a.bias = nullptr; a.ref = nullptr; a.slope = 0.2f; a.gain = 1.f;
  for (int i = 0; i < 3; ++i) launch<MODE_FWD>(a, dwp, dU, C, K, 0);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch<MODE_FWD>(a, dwp, dU, C, K, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  g_filter = false;
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch<MODE_FWD>(a, dwp, dU, C, K, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms_k; (void)hipEventElapsedTime(&ms_k, e0, e1); ms_k /= reps;
  g_filter = true;
  const double gf = 2.0 * N * H * W * (double)C * K * 9 * 1e-9;
  const int items = a.NP * a.NKB;
  printf("time N=%-5d H=%-4d W=%-4d C=%-4d K=%-4d items %-5d (%.2f rounds)  %8.1f us (filter + kernel; kernel alone %7.1f)  nominal %6.1f TF/s  issued %6.1f TF/s (%.3f of 157.3)\n",
         N, H, W, C, K, items, items / 256.0, ms * 1e3, ms_k * 1e3, gf / ms, gf * 0.25 / ms, gf * 0.25 / ms / 157.3);
  (void)hipFree(din); (void)hipFree(dwp); (void)hipFree(dout); (void)hipFree(dU);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  if (argc > 2 && argv[2][0] == 'z') g_data = 2;
  if (!(argc > 3 && argv[3][0] == 's')) {
    double worst = 0;
    worst = std::max(worst, check(0, 3, 16, 16, 32, 64));
    worst = std::max(worst, check(1, 3, 16, 16, 64, 32));
    worst = std::max(worst, check(0, 2, 32, 64, 32, 64));
    worst = std::max(worst, check(1, 1, 64, 32, 128, 32));
    worst = std::max(worst, check(0, 17, 16, 16, 32, 128));
    worst = std::max(worst, check(0, 19, 8, 8, 32, 64));
    worst = std::max(worst, check(1, 13, 8, 8, 64, 32));
    if (worst >= 2e-5) { printf("FAILED\n"); return 1; }
  }
  timeit(1536, 16, 16, 128, 128, reps);
  timeit(1536, 8, 8, 256, 256, reps);
  timeit(48, 128, 128, 128, 128, reps);
  timeit(48, 256, 256, 64, 64, reps);
  timeit(48, 64, 64, 256, 256, reps);
  timeit(48, 32, 32, 512, 512, reps);
  timeit(16, 128, 128, 128, 128, reps);
  timeit(16, 64, 64, 256, 256, reps);
  timeit(1536, 16, 16, 128, 128, reps);
  return 0;
}
