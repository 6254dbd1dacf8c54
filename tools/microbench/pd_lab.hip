// Lab for the persistent decode step (csrc/pdecode.hip): synthetic bf16 weights of a GPT size, M rows at position `pos`, the kernel
// replayed `reps` times with in-kernel time stamps of layer 1 (-DVLG_PD_PROF).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVLG_PD_PROF -I video-llamagen_amd/csrc -I include tools/microbench/pd_lab.hip \
//         video-llamagen_amd/csrc/core.hip -o gpurun_out/pd_lab && gpurun_out/pd_lab XL 16 600 [reps] [form: 1 = pdecode.hip, 2 = pdecode2.hip]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../video-llamagen_amd/csrc/pdecode.hip"
#include "pdecode2_lab.hip"

using namespace vlg;

#define CK(e)                                                                      \
  do {                                                                             \
    hipError_t _e = (e);                                                           \
    if (_e != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void fill_kernel(uint16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = f32_to_bf16(((h & 0xffff) / 65536.0f - 0.5f) * scale);
  }
}
static bf16* alloc_fill(size_t n, uint32_t seed, float scale) {
  bf16* p;
  CK(hipMalloc(&p, n * 2));
  fill_kernel<<<1024, 256>>>((uint16_t*)p, n, seed, scale);
  return p;
}
__global__ void set_state_k(StepState* s, int pos, int step) {
  s->pos = pos;
  s->step = step;
}

int main(int argc, char** argv) {
  const char* model = argc > 1 ? argv[1] : "XL";
  const int M = argc > 2 ? atoi(argv[2]) : 16;
  const int pos = argc > 3 ? atoi(argv[3]) : 600;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  const int form = argc > 5 ? atoi(argv[5]) : 1;
  int D = 1280, H = 20, L = 36;
  if (!strcmp(model, "L")) D = 1024, H = 16, L = 24;
  if (!strcmp(model, "B")) D = 768, H = 12, L = 12;
  const int hd = 64, S = ((pos + reps + 8) / 8) * 8;
  int F = 4 * D;
  F = 2 * F / 3;
  F = (F + 255) / 256 * 256;
  std::vector<PdLayer> hl(L);
  for (int l = 0; l < L; ++l) {
    hl[l].wqkv = alloc_fill((size_t)3 * D * D, 11 * l + 1, 0.04f);
    hl[l].wo = alloc_fill((size_t)D * D, 11 * l + 2, 0.04f);
    hl[l].w13 = alloc_fill((size_t)2 * F * D, 11 * l + 3, 0.04f);
    hl[l].w2 = alloc_fill((size_t)D * F, 11 * l + 4, 0.04f);
    hl[l].norm1 = alloc_fill(D, 11 * l + 5, 2.0f);
    hl[l].norm2 = alloc_fill(D, 11 * l + 6, 2.0f);
  }
  PdLayer* dl;
  CK(hipMalloc(&dl, L * sizeof(PdLayer)));
  CK(hipMemcpy(dl, hl.data(), L * sizeof(PdLayer), hipMemcpyHostToDevice));
  bf16* x0 = alloc_fill((size_t)M * D, 777, 2.0f);
  bf16* x = alloc_fill((size_t)M * D, 777, 2.0f);
  const size_t kvl = (size_t)M * H * S * hd;
  bf16* kc = alloc_fill(kvl * L, 781, 1.0f);
  bf16* vc = alloc_fill(kvl * L, 782, 1.0f);
  float* freqs;
  CK(hipMalloc(&freqs, (size_t)S * hd * 4));
  CK(hipMemset(freqs, 0, (size_t)S * hd * 4));
  StepState* state;
  CK(hipMalloc(&state, sizeof(StepState)));
  unsigned* fault;
  CK(hipHostMalloc((void**)&fault, 64, hipHostMallocMapped));
  memset(fault, 0, 64);
  unsigned* fault_dev;
  CK(hipHostGetDevicePointer((void**)&fault_dev, fault, 0));
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  if (form == 2 ? !pd2_ok<bf16>(M, D, H, hd, F, cus) : !pd_ok<bf16>(M, D, H, hd, F, S, cus)) {
    fprintf(stderr, "shape not covered\n");
    return 1;
  }
  void* xbuf;
  const size_t xbytes = form == 2 ? pd2_xbuf_bytes(M, D, H, hd, F, 2, cus) : pd_xbuf_bytes(M, D, H, hd, F, 2);
  CK(hipMalloc(&xbuf, xbytes));
  CK(hipMemset(xbuf, 0, xbytes));
  unsigned long long* prof;
  CK(hipMalloc(&prof, (size_t)cus * 32 * 8));
  CK(hipMemset(prof, 0, (size_t)cus * 32 * 8));
  PdArgs a{};
  a.layers = dl; a.x = x; a.kc = kc; a.vc = vc; a.kv_lstride = kvl; a.freqs = freqs; a.state = state; a.mask = nullptr; a.Bmask = 1; a.Tc = 1;
  a.xbuf = xbuf; a.fault = fault_dev; a.spin_max = 200000;
  a.L = L; a.M = M; a.D = D; a.H = H; a.hd = hd; a.F = F; a.S = S; a.eps = 1e-5f;
  a.prof = prof;
  a.fm = 1;   // the product streams fragment-major weights (random fill: the layout does not matter for the values here)
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float tot = 0;
  for (int i = 0; i < reps + 3; ++i) {
    CK(hipMemcpyAsync(x, x0, (size_t)M * D * 2, hipMemcpyDeviceToDevice, st));
    set_state_k<<<1, 1, 0, st>>>(state, pos + i, 1 + i);
    CK(hipEventRecord(e0, st));
    if ((form == 2 ? pd2_layers<bf16>(a, st) : pd_layers<bf16>(a, st)) != VLG_OK) {
      fprintf(stderr, "launch failed: %s\n", vlg_last_error());
      return 1;
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 3) tot += ms;
    if (*fault) {
      fprintf(stderr, "fault 0x%08x at rep %d\n", *fault, i);
      return 2;
    }
  }
  printf("%s M=%d pos=%d form %d: %.1f us per step, %.2f us per layer\n", model, M, pos, form, tot / reps * 1e3, tot / reps * 1e3 / L);
  if (form == 2) {
    const Pd2Geom geo(M, D, H, hd, 2, cus);
    printf("  geometry: row groups of %d, %d shares per (row group, head) = %d items; reducer chunks of %d columns = %d units; %d MLP slices\n", geo.rg, geo.gs,
           geo.nitems, geo.cw, geo.nunits, F / 16);
    std::vector<unsigned long long> p2((size_t)cus * 32);
    CK(hipMemcpy(p2.data(), prof, p2.size() * 8, hipMemcpyDeviceToHost));
    const char* n2[13] = {"layer start", "x swept", "qkv mfma+red", "qkv published", "q ready", "kv streamed", "shares swept", "wo partials out", "h reduced+out",
                          "h swept", "w13 mfma+red", "w2 partials out", "x reduced+out"};
    for (int w : {0, 1, 79, 100, 230, 250, 255}) {
      if (w >= cus) continue;
      const unsigned long long* t = p2.data() + (size_t)w * 32;
      printf("wg %3d:", w);
      unsigned long long prev = t[0];
      for (int i = 0; i < 13; ++i) {
        if (!t[i]) continue;
        printf(" [%d %s +%.2f]", i, n2[i], (double)(t[i] - prev) / 100.0);
        prev = t[i];
      }
      printf("  total %.2f us\n", (double)(t[12] - t[0]) / 100.0);
      if (t[13]) printf("        fine: q ready -> K / V loop done %.2f, -> waves merged in LDS %.2f, -> partial out + barrier %.2f\n", (double)(t[13] - t[4]) / 100.0,
                        (double)(t[14] - t[13]) / 100.0, (double)(t[5] - t[14]) / 100.0);
    }
    return 0;
  }
  std::vector<unsigned long long> p((size_t)cus * 32);
  CK(hipMemcpy(p.data(), prof, p.size() * 8, hipMemcpyDeviceToHost));
  const char* names[15] = {"layer start", "x swept", "qkv mfma+red", "qkv published", "att q ready", "att streamed", "att done", "ao swept", "wo published",
                           "h swept", "w13 mfma+red", "g published", "g chunk0 swept", "g chunk1 swept", "layer end"};
  for (int w : {0, 1, 79, 100, 230, 255}) {
    if (w >= cus) continue;
    const unsigned long long* t = p.data() + (size_t)w * 32;
    printf("wg %3d:", w);
    unsigned long long prev = t[0];
    for (int i = 0; i < 15; ++i) {
      if (!t[i]) continue;
      printf(" [%d %s +%.2f]", i, names[i], (double)(t[i] - prev) / 100.0);
      prev = t[i];
    }
    printf("  total %.2f us\n", (double)(t[14] - t[0]) / 100.0);
    printf("        fine: wo gemm %.2f, +red/prefetch %.2f, +barrier %.2f | qkv norm %.2f, +barrier %.2f (us after stamps 7 / 1)\n", (double)(t[15] - t[7]) / 100.0,
           (double)(t[16] - t[15]) / 100.0, (double)(t[17] - t[16]) / 100.0, (double)(t[18] - t[1]) / 100.0, (double)(t[19] - t[18]) / 100.0);
  }
  return 0;
}
