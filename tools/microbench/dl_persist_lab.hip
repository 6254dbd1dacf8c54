// Timing lab for the persistent DiffLoss sampler (csrc/diffloss_persist.hip) on synthetic weights: whole-launch time and in-kernel
// time stamps of one reverse step (workgroup 0).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVLG_DP_PROF -I video-llamagen_amd/csrc -I include tools/microbench/dl_persist_lab.hip \
//         video-llamagen_amd/csrc/core.hip -o tools/microbench/bin/dl_persist_lab && tools/microbench/bin/dl_persist_lab [B] [W] [S] [rows per group]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../video-llamagen_amd/csrc/diffloss_persist.hip"

using namespace vlg;

#define CK(e)                                                                   \
  do {                                                                          \
    hipError_t _e = (e);                                                        \
    if (_e != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed, float scale, float offs) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = f32_to_bf16(((h & 0xffff) / 65536.0f - 0.5f) * scale + offs);
  }
}
static bf16* alloc_fill(size_t n, uint32_t seed, float scale, float offs = 0.f) {
  bf16* p;
  CK(hipMalloc(&p, n * 2));
  fill_kernel<<<256, 256>>>((uint16_t*)p, n, seed, scale, offs);
  return p;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, W = argc > 2 ? atoi(argv[2]) : 1024, S = argc > 3 ? atoi(argv[3]) : 100;
  const int C = 8, depth = 3, MR = (3 * depth + 2) * W;
  if (!dl_persist_ok<bf16>(B, W, C, depth, argc > 4 ? atoi(argv[4]) : 0)) {
    printf("shape not covered\n");
    return 1;
  }
  DlPersist p{};
  for (int b = 0; b < depth; ++b) {
    p.ln_w[b] = alloc_fill(W, 100 + b, 0.1f, 1.0f);
    p.ln_b[b] = alloc_fill(W, 200 + b, 0.1f);
    p.w0[b] = alloc_fill((size_t)W * W, 300 + b, 0.06f);
    p.b0[b] = alloc_fill(W, 400 + b, 0.1f);
    p.w2[b] = alloc_fill((size_t)W * W, 500 + b, 0.06f);
    p.b2[b] = alloc_fill(W, 600 + b, 0.1f);
  }
  p.wf = alloc_fill((size_t)2 * C * W, 700, 0.06f);
  p.bf = alloc_fill(2 * C, 701, 0.1f);
  p.wip = alloc_fill((size_t)W * C, 702, 0.5f);
  p.bip = alloc_fill(W, 703, 0.1f);
  p.mod_all = alloc_fill((size_t)S * B * MR, 704, 0.2f);
  std::vector<DdpmCoef> coef(S);
  for (int i = 0; i < S; ++i) coef[i] = DdpmCoef{1.01f, 0.1f, 0.05f, 0.94f, -6.f, -4.f, i > 0 ? 1 : 0};
  DdpmCoef* dcoef;
  CK(hipMalloc(&dcoef, S * sizeof(DdpmCoef)));
  CK(hipMemcpy(dcoef, coef.data(), S * sizeof(DdpmCoef), hipMemcpyHostToDevice));
  p.coef = dcoef;
  StepState hs{0, 0}, *dstate;
  CK(hipMalloc(&dstate, sizeof(StepState)));
  CK(hipMemcpy(dstate, &hs, sizeof(hs), hipMemcpyHostToDevice));
  p.state = dstate;
  CK(hipMalloc(&p.xbuf, dl_persist_xbuf_bytes(B, W, 2)));
  CK(hipMalloc((void**)&p.cur, B * C * 4));
  CK(hipMalloc((void**)&p.out_lat, B * C * 4));
  CK(hipMalloc((void**)&p.prof, 64 * 8));
  CK(hipMemset(p.prof, 0, 64 * 8));
  p.depth = depth; p.W = W; p.C = C; p.S = S; p.B = B; p.MR = MR; p.N = 1; p.b_off = 0; p.B_total = B;
  p.temperature = 1.0f;
  p.spin_max = 1 << 20;
  p.rows = argc > 4 ? atoi(argv[4]) : 0;   // rows per group: 0 = chosen by the batch (4 up to 32 rows at W 1024, 8 up to 64)
  p.seed = 1234;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w)
    if (dl_persist<bf16>(p, st) != 0) return 2;
  CK(hipStreamSynchronize(st));
  const int reps = 10;
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) dl_persist<bf16>(p, st);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> out(B * C);
  CK(hipMemcpy(out.data(), p.cur, B * C * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (float v : out) bad += !(v == v);
  printf("{\"B\": %d, \"W\": %d, \"S\": %d, \"ms_per_token\": %.4f, \"us_per_reverse_step\": %.2f, \"nan_outputs\": %d, \"x0\": %.4f", B, W, S,
         ms / reps, ms / reps * 1e3 / S, bad, out[0]);
  unsigned long long prof[64];
  CK(hipMemcpy(prof, p.prof, sizeof(prof), hipMemcpyDeviceToHost));
  const char* names[12] = {"step_start", "ln0", "gemm_mlp0", "publish0", "collect0", "gemm_mlp2", "publish2", "collect2", "blocks_done",
                           "final_gemm", "ddpm", "input_proj"};
  printf(", \"stamps_us_since_prev\": {");
  for (int i = 1; i < 12; ++i) {
    const unsigned long long prev = prof[i == 8 ? 0 : i - 1];
    printf("%s\"%s%s\": %.2f", i > 1 ? ", " : "", names[i], i == 8 ? "_since_step_start" : "", (double)(prof[i] - prev) / 100.0);
  }
  printf("}, \"block1\": {\"ln_wave0\": %.2f, \"ln_barrier\": %.2f, \"total\": %.2f}, \"block0_total\": %.2f, \"block2_total\": %.2f}\n",
         (double)(prof[13] - prof[12]) / 100.0, (double)(prof[14] - prof[13]) / 100.0, (double)(prof[15] - prof[12]) / 100.0,
         (double)(prof[7] - prof[0]) / 100.0, (double)(prof[8] - prof[15]) / 100.0);
  return bad != 0;
}
