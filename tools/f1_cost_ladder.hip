// Where do the 12.5 us of an F launch go?  A ladder: one launch per iteration (1024 workgroups x 256 threads, block -> XCD mapping of the engine), starting
// from the bare vector traffic of an F launch (8 vectors on a 300-column window read, 100 own columns x 7 + a 300-column replica slice written, four
// workgroup barriers) and adding the engine's other ingredients one at a time (template bit mask):
//   1  the scalar fold: 3 x 1024 partials of the previous launch read by every workgroup, block reduction, partials written at the end
//   2  the phase record: 16 words read with scalar loads at the head, written by workgroup 0 at the end
//   4  the block's matrix stream: 12 KB per block (values + index words) into LDS, consumed by the product phase
//   8  pointers / P: row pointers + rho, column pointers, P row pointers, one P entry (value + column) per lane
//  16  the LDS phases at their real size: 1000 products, row sums, 1000 transposed products, column sums
//  32  (with 1) the fold by wave 3 alone, while waves 0-2 request the window: the two round trips to memory overlap
// 256  the fold as 64-bit INTEGER atomics (round 6): every workgroup adds its three fixed-point partials to three words (order-independent, hence
//      deterministic), the next launch reads those three words with one scalar load instead of 24 KB of partials; three sets, the third zeroed by workgroup 0
// 512  ... the same spread over 8 word triples (one per XCD = blockIdx.x & 7): eight times fewer adds per address, 24 words read at the head
//   hipcc --offload-arch=gfx950 -O3 tools/f1_cost_ladder.hip -o /tmp/ladder && timeout 120 /tmp/ladder
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int C = 100, W = 300, NV = 8, G = 1024, ROWS = 200, ENT = 1000;
template <int F, int Q>
__global__ __launch_bounds__(256 * Q) void k_f(double *vec, size_t ns, int k, double *part, int *rec, const double *stream, const int *aux, double *out) {
  __shared__ double win_[Q][512], prod_[Q][1024], tv_[Q][512], red[64];
  __shared__ double sval_[Q][1024]; __shared__ unsigned sent_[Q][1024];
  const int sub = threadIdx.x >> 8;                          // Q sub-blocks of 256 threads per workgroup, one row block each
  double *win = win_[sub], *prod = prod_[sub], *tv = tv_[sub], *sval = sval_[sub]; unsigned *sent = sent_[sub];
  const int per = (G + 7) >> 3;
  const int vb = (int)blockIdx.x * Q + sub;                 // virtual 256-thread block
  const int b = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) * Q + sub;
  const int tid = threadIdx.x & 255;
  const int c0 = b * C, g0 = max(0, c0 - (W - C) / 2);
  const int par = k & 1;
  double alpha = 1e-3;
  int phase = 1;
  if (F & 2) { const int *R = rec + (par ? 16 : 0); phase = __builtin_nontemporal_load(R + 1) + __builtin_nontemporal_load(R + 5) * 0; }
  double accp[2] = {0.0, 0.0}; bool pre = false;
  if ((F & 1) && (F & 32) && Q == 1) {                                 // the fold by wave 3 ALONE while waves 0-2 request the window (192 lanes x 2 columns)
    pre = true;
    const double *src = vec + (size_t)((k - 1) & 1) * NV * ns;
    if (tid >= 192) {
      const double *pp = part + (size_t)(par ^ 1) * 3 * G; const int l = tid - 192;
      double a = 0, c = 0, m = 0;
      double va[G / 64], vc[G / 64], vm[G / 64];             // coalesced: lane l takes elements l, l + 64, ... (all requests out before the first add)
#pragma unroll
      for (int q = 0; q < G / 64; q++) { va[q] = pp[q * 64 + l]; vc[q] = pp[G + q * 64 + l]; vm[q] = pp[2 * G + q * 64 + l]; }
#pragma unroll
      for (int q = 0; q < G / 64; q++) { a += va[q]; c += vc[q]; m = fmax(m, vm[q]); }
      for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); m = fmax(m, __shfl_xor(m, o)); }
      if (l == 0) { red[0] = a; red[1] = c; red[2] = m; }
    } else {
      for (int u = 0; u < 2; u++) { const int e = tid + u * 192; if (e < W) { const int c = g0 + e; double w = 0; for (int v = 0; v < NV; v++) w += src[(size_t)v * ns + c]; accp[u] = w; } }
    }
    __syncthreads();
    alpha = 1e-3 + 1e-12 * red[0] / (1.0 + fabs(red[1]) + red[2]);
  } else if (F & 1) {                                               // fold of the previous launch's partials (one slot per WORKGROUP: G / Q of them)
    const double *pp = part + (size_t)(par ^ 1) * 3 * G;
    double a = 0, c = 0, m = 0;
    if (Q == 1) { for (int q = 0; q < G / 256; q++) { a += pp[tid * (G / 256) + q]; c += pp[G + tid * (G / 256) + q]; m = fmax(m, pp[2 * G + tid * (G / 256) + q]); } }
    else if (threadIdx.x < G / Q) { a = pp[threadIdx.x]; c = pp[G + threadIdx.x]; m = pp[2 * G + threadIdx.x]; }
    for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); m = fmax(m, __shfl_xor(m, o)); }
    { const int wv = threadIdx.x >> 6; if ((threadIdx.x & 63) == 0) { red[wv] = a; red[16 + wv] = c; red[32 + wv] = m; } }
    __syncthreads();
    a = 0; c = 0; m = 0; for (int w = 0; w < 4 * Q; w++) { a += red[w]; c += red[16 + w]; m = fmax(m, red[32 + w]); }
    __syncthreads();
    alpha = 1e-3 + 1e-12 * a / (1.0 + fabs(c) + m);
  }
  if (F & 64) {                                              // one dependent 8-byte read per lane at the head (2 KB per workgroup, written by the previous launch), no reduction
    const double t = part[(size_t)(par ^ 1) * 3 * G + ((tid * 4 + (int)blockIdx.x) & (G - 1))];
    if (F & 128) { red[tid >> 6] = t; __syncthreads(); alpha += 1e-12 * red[0]; __syncthreads(); }      // ... published through LDS behind a barrier
    else alpha += 1e-12 * t;
  }
  if (F & 256) {                                             // the three words the previous launch accumulated (spread: 8 triples)
    const unsigned long long *acw = reinterpret_cast<const unsigned long long *>(part + 2 * 3 * G) + (size_t)((k + 2) % 3) * 32;
    long long a = 0, c = 0; unsigned long long m = 0;
    if (F & 512) { for (int x = 0; x < 8; x++) { a += (long long)acw[4 * x]; c += (long long)acw[4 * x + 1]; m = max(m, acw[4 * x + 2]); } }
    else { a = (long long)acw[0]; c = (long long)acw[1]; m = acw[2]; }
    alpha = 1e-3 + 1e-12 * (double)a / (1.0 + fabs((double)c) + (double)m);
  }
  if (phase == 0) return;
  // ---- loads
  const double *src = vec + (size_t)((k - 1) & 1) * NV * ns; double *dst = vec + (size_t)(k & 1) * NV * ns;
  double pv = 0; int pc = 0, rp0 = 0, rp1 = 0, cp[4] = {0, 0, 0, 0}, pp0 = 0, pp1 = 0; double rho = 1.0;
  if (F & 8) { const int *ax = aux + (size_t)b * 2048; pv = src[(size_t)7 * ns + c0 + (tid % C)] * 1e-3; pc = ax[tid]; rp0 = ax[256 + min(tid, ROWS)]; rp1 = ax[257 + min(tid, ROWS)]; rho = src[(size_t)6 * ns + c0 + tid % C];
               cp[0] = ax[600 + min(tid, W)]; cp[1] = ax[601 + min(tid, W)]; cp[2] = ax[600 + min(tid + 256, W)]; cp[3] = ax[601 + min(tid + 256, W)]; pp0 = ax[1000 + min(tid, C)]; pp1 = ax[1001 + min(tid, C)]; }
  double acc[2] = {0.0, 0.0};
  if (pre) { if (tid < 192) for (int u = 0; u < 2; u++) { const int e = tid + u * 192; if (e < W) { acc[u] = accp[u] * alpha; win[e] = acc[u]; } } }
  else for (int u = 0; u < 2; u++) { const int e = tid + u * 256; if (e < W) { const int c = g0 + e; double w = 0; for (int v = 0; v < NV; v++) w += src[(size_t)v * ns + c]; acc[u] = w * alpha; win[e] = acc[u]; } }
  double4 sv[3];
  if (F & 4) { const double4 *st = reinterpret_cast<const double4 *>(stream + (size_t)b * 1536); for (int q = 0; q < 3; q++) sv[q] = st[tid + q * 256 < 384 ? tid + q * 256 : 0]; }
  if (F & 4) { for (int q = 0; q < 3; q++) { const int i = tid + q * 256; if (i < 256) { sval[4 * i] = sv[q].x; sval[4 * i + 1] = sv[q].y; sval[4 * i + 2] = sv[q].z; sval[4 * i + 3] = sv[q].w; } else if (i < 384) { sent[8 * (i - 256)] = (unsigned)__double_as_longlong(sv[q].x); } } }
  __syncthreads();
  double s = 0.0;
  if (F & 16) {
    double vw[4]; unsigned en[4];
    for (int u = 0; u < 4; u++) { const int e = min(tid + u * 256, ENT - 1); vw[u] = (F & 4) ? sval[e] + 1.0 : 1.0 + e * 1e-6; en[u] = (unsigned)((e * 7) % W) | ((unsigned)(e / 5) << 9) | ((unsigned)((e * 13) % ENT) << 18); }
    for (int u = 0; u < 4; u++) prod[min(tid + u * 256, ENT - 1)] = vw[u] * win[en[u] & 511];
    __syncthreads();
    if (tid < ROWS) { double a = 0; for (int j = 0; j < 5; j++) a += prod[tid * 5 + j]; tv[tid] = rho * a; s += a; }
    __syncthreads();
    for (int u = 0; u < 4; u++) prod[en[u] >> 18] = vw[u] * tv[(en[u] >> 9) & 511];
    __syncthreads();
    for (int u = 0; u < 2; u++) { const int e = tid + u * 256; if (e < W) { double a = 0; for (int j = 0; j < 3; j++) a += prod[(e * 3 + j) % ENT]; acc[u] += 1e-9 * a; } }
  } else {
    for (int j = 0; j < 4; j++) s += win[(tid + j * 37) % W];
    __syncthreads(); tv[tid] = s; __syncthreads();
    s += tv[(tid * 3) % 256]; __syncthreads();
  }
  s += pv * 1e-9 + (pc + rp0 + rp1 + cp[0] + cp[1] + cp[2] + cp[3] + pp0 + pp1) * 1e-12;
  if (tid < C) for (int v = 0; v < NV - 1; v++) dst[(size_t)v * ns + c0 + tid] = 1e-3 * s + v;
  for (int u = 0; u < 2; u++) { const int e = pre ? (tid < 192 ? tid + u * 192 : W) : tid + u * 256; if (e < W && (b % 4 == 0 || (g0 + e >= c0 && g0 + e < c0 + C))) dst[(size_t)(NV - 1) * ns + g0 + e] = 1e-3 * acc[u]; }
  if (F & 1) {                                               // partials of this launch
    double a = s, c = s * 0.5, m = fabs(s);
    for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); m = fmax(m, __shfl_xor(m, o)); }
    __syncthreads();
    { const int wv = threadIdx.x >> 6; if ((threadIdx.x & 63) == 0) { red[wv] = a; red[16 + wv] = c; red[32 + wv] = m; } }
    __syncthreads();
    if (threadIdx.x == 0) { double ra = 0, rc = 0, rm = 0; for (int w = 0; w < 4 * Q; w++) { ra += red[w]; rc += red[16 + w]; rm = fmax(rm, red[32 + w]); } double *pw = part + (size_t)par * 3 * G; pw[blockIdx.x] = ra; pw[G + blockIdx.x] = rc; pw[2 * G + blockIdx.x] = rm; }
  }
  if (F & 256) {
    double a = s, c = s * 0.5, m = fabs(s);
    for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); m = fmax(m, __shfl_xor(m, o)); }
    __syncthreads();
    { const int wv = threadIdx.x >> 6; if ((threadIdx.x & 63) == 0) { red[wv] = a; red[16 + wv] = c; red[32 + wv] = m; } }
    __syncthreads();
    unsigned long long *acb = reinterpret_cast<unsigned long long *>(part + 2 * 3 * G);
    if (threadIdx.x < 3) {
      double r = 0; for (int w = 0; w < 4 * Q; w++) r = threadIdx.x == 2 ? fmax(r, red[32 + w]) : r + red[16 * threadIdx.x + w];
      unsigned long long *dstw = acb + (size_t)(k % 3) * 32 + ((F & 512) ? 4 * (blockIdx.x & 7) : 0) + threadIdx.x;
      const long long fx = (long long)(r * 1048576.0);
      if (threadIdx.x == 2) atomicMax(dstw, (unsigned long long)fx); else atomicAdd(dstw, (unsigned long long)fx);
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) acb[(size_t)((k + 1) % 3) * 32 + threadIdx.x] = 0ull;
  }
  if ((F & 64) && tid == 0) part[(size_t)par * 3 * G + blockIdx.x * Q + sub] = s;      // (one 8-byte store per block at the end)
  if ((F & 2) && blockIdx.x == 0 && threadIdx.x < 16) rec[(par ? 0 : 16) + threadIdx.x] = threadIdx.x == 1 ? 1 : k;
  if (tid == 0 && s == 123.456) out[blockIdx.x] = s;
}
// Taller row blocks: 512 workgroups x 512 threads, one block of 400 rows / 2000 entries per workgroup -- own columns 200, window 400 (the band's 200 +
// the rows' 200): every column's 8 values are gathered by 2.0 blocks instead of 2.9, half as many partial slots.  Everything of the ladder switched on.
__global__ __launch_bounds__(512) void k_tall(double *vec, size_t ns, int k, double *part, int *rec, const double *stream, const int *aux, double *out) {
  constexpr int G2 = G / 2, C2 = 2 * C, W2 = 400, ROWS2 = 2 * ROWS, ENT2 = 2 * ENT;
  __shared__ double win[1024], prod[2048], tv[1024], red[64];
  __shared__ double sval[2048]; __shared__ unsigned sent[2048];
  const int per = (G2 + 7) >> 3;
  const int b = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  const int tid = threadIdx.x;
  const int c0 = b * C2, g0 = max(0, c0 - (W2 - C2) / 2);
  const int par = k & 1;
  double alpha = 1e-3;
  const int *R = rec + (par ? 16 : 0); const int phase = __builtin_nontemporal_load(R + 1);
  { const double *pp = part + (size_t)(par ^ 1) * 3 * G;
    double a = pp[tid], c = pp[G + tid], m = pp[2 * G + tid];
    for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); m = fmax(m, __shfl_xor(m, o)); }
    const int wv = tid >> 6; if ((tid & 63) == 0) { red[wv] = a; red[16 + wv] = c; red[32 + wv] = m; }
    __syncthreads();
    a = 0; c = 0; m = 0; for (int w = 0; w < 8; w++) { a += red[w]; c += red[16 + w]; m = fmax(m, red[32 + w]); }
    __syncthreads();
    alpha = 1e-3 + 1e-12 * a / (1.0 + fabs(c) + m); }
  if (phase == 0) return;
  const double *src = vec + (size_t)((k - 1) & 1) * NV * ns; double *dst = vec + (size_t)(k & 1) * NV * ns;
  const int *ax = aux + (size_t)b * 4096;
  const double pv = src[(size_t)7 * ns + c0 + (tid % C2)] * 1e-3; const int pc = ax[tid], rp0 = ax[512 + min(tid, ROWS2)], rp1 = ax[513 + min(tid, ROWS2)]; const double rho = src[(size_t)6 * ns + c0 + tid % C2];
  const int cp0 = ax[1200 + min(tid, W2)], cp1 = ax[1201 + min(tid, W2)], pp0 = ax[2000 + min(tid, C2)], pp1 = ax[2001 + min(tid, C2)];
  double acc = 0.0;
  if (tid < W2) { const int c = g0 + tid; double w = 0; for (int v = 0; v < NV; v++) w += src[(size_t)v * ns + c]; acc = w * alpha; win[tid] = acc; }
  { const double4 *st = reinterpret_cast<const double4 *>(stream + (size_t)b * 3072); double4 sv[2]; for (int q = 0; q < 2; q++) sv[q] = st[tid + q * 512 < 768 ? tid + q * 512 : 0];
    for (int q = 0; q < 2; q++) { const int i = tid + q * 512; if (i < 512) { sval[4 * i] = sv[q].x; sval[4 * i + 1] = sv[q].y; sval[4 * i + 2] = sv[q].z; sval[4 * i + 3] = sv[q].w; } else if (i < 768) sent[8 * (i - 512)] = (unsigned)__double_as_longlong(sv[q].x); } }
  __syncthreads();
  double s = 0.0;
  double vw[4]; unsigned en[4];
  for (int u = 0; u < 4; u++) { const int e = min(tid + u * 512, ENT2 - 1); vw[u] = sval[e] + 1.0; en[u] = (unsigned)((e * 7) % W2) | ((unsigned)(e / 5) << 10) | ((unsigned)((e * 13) % ENT2) << 20); }
  for (int u = 0; u < 4; u++) prod[min(tid + u * 512, ENT2 - 1)] = vw[u] * win[en[u] & 1023];
  __syncthreads();
  if (tid < ROWS2) { double a = 0; for (int j = 0; j < 5; j++) a += prod[tid * 5 + j]; tv[tid] = rho * a; s += a; }
  __syncthreads();
  for (int u = 0; u < 4; u++) prod[en[u] >> 20] = vw[u] * tv[(en[u] >> 10) & 1023];
  __syncthreads();
  if (tid < W2) { double a = 0; for (int j = 0; j < 5; j++) a += prod[(tid * 5 + j) % ENT2]; acc += 1e-9 * a; }
  s += pv * 1e-9 + (pc + rp0 + rp1 + cp0 + cp1 + pp0 + pp1) * 1e-12;
  if (tid < C2) for (int v = 0; v < NV - 1; v++) dst[(size_t)v * ns + c0 + tid] = 1e-3 * s + v;
  if (tid < W2 && (b % 2 == 0 || (g0 + tid >= c0 && g0 + tid < c0 + C2))) dst[(size_t)(NV - 1) * ns + g0 + tid] = 1e-3 * acc;
  { double a = s, c = s * 0.5, m = fabs(s);
    for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); c += __shfl_xor(c, o); m = fmax(m, __shfl_xor(m, o)); }
    __syncthreads();
    const int wv = tid >> 6; if ((tid & 63) == 0) { red[wv] = a; red[16 + wv] = c; red[32 + wv] = m; }
    __syncthreads();
    if (tid == 0) { double ra = 0, rc = 0, rm = 0; for (int w = 0; w < 8; w++) { ra += red[w]; rc += red[16 + w]; rm = fmax(rm, red[32 + w]); } double *pw = part + (size_t)par * 3 * G; pw[blockIdx.x] = ra; pw[G + blockIdx.x] = rc; pw[2 * G + blockIdx.x] = rm; } }
  if (blockIdx.x == 0 && tid < 16) rec[(par ? 0 : 16) + tid] = tid == 1 ? 1 : k;
  if (tid == 0 && s == 123.456) out[blockIdx.x] = s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int F, int Q = 1> int run(const char *what, double *vec, size_t ns, double *part, int *rec, double *stream, int *aux, double *out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000; float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0));
    for (int k = 1; k <= iters; k++) hipLaunchKernelGGL((k_f<F, Q>), dim3(G / Q), dim3(256 * Q), 0, 0, vec, ns, k, part, rec, stream, aux, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::printf("%-100s %6.2f us per launch\n", what, 1e3 * ms / iters);
  return 0;
}
int main() {
  const size_t ns = (size_t)G * C + 512;
  double *vec, *part, *stream, *out; int *rec, *aux;
  CK(hipMalloc(&vec, 8 * 2 * NV * ns)); CK(hipMemset(vec, 0, 8 * 2 * NV * ns)); CK(hipMalloc(&part, 8 * (2 * 3 * G + 96))); CK(hipMemset(part, 0, 8 * (2 * 3 * G + 96)));
  CK(hipMalloc(&stream, 8 * (size_t)G * 1536)); CK(hipMemset(stream, 0, 8 * (size_t)G * 1536)); CK(hipMalloc(&out, 8 * G)); CK(hipMalloc(&rec, 4 * 32)); CK(hipMalloc(&aux, 4 * (size_t)G * 2048)); CK(hipMemset(aux, 0, 4 * (size_t)G * 2048));
  int one[32]; for (int i = 0; i < 32; i++) one[i] = 1; CK(hipMemcpy(rec, one, sizeof(one), hipMemcpyHostToDevice));
  if (run<0>("vector traffic of an F launch + four barriers", vec, ns, part, rec, stream, aux, out)) return 1;
  run<1>("+ scalar fold (3 x 1024 partials read by every workgroup, partials written)", vec, ns, part, rec, stream, aux, out);
  run<3>("+ phase record (scalar loads at the head, written by workgroup 0)", vec, ns, part, rec, stream, aux, out);
  run<7>("+ matrix stream (12 KB per block through LDS)", vec, ns, part, rec, stream, aux, out);
  run<15>("+ row / column / P pointers, rho, one P entry per lane", vec, ns, part, rec, stream, aux, out);
  run<31>("+ LDS phases at their real size (1000 products, row sums, transposed products, column sums)", vec, ns, part, rec, stream, aux, out);
  run<30>("the same without the scalar fold", vec, ns, part, rec, stream, aux, out);
  run<29>("the same without the phase record", vec, ns, part, rec, stream, aux, out);
  run<27>("the same without the matrix stream", vec, ns, part, rec, stream, aux, out);
  run<30 | 256>("everything, the fold as three 64-bit integer atomics per workgroup + one scalar read of three words at the head", vec, ns, part, rec, stream, aux, out);
  run<30 | 256 | 512>("... the atomics spread over eight word triples (by blockIdx & 7)", vec, ns, part, rec, stream, aux, out);
  run<256>("vector traffic + the atomic fold", vec, ns, part, rec, stream, aux, out);
  run<63>("everything, the fold by wave 3 alone while waves 0-2 request the window", vec, ns, part, rec, stream, aux, out);
  run<33>("vector traffic + the fold by wave 3 alone under the window requests", vec, ns, part, rec, stream, aux, out);
  run<64>("vector traffic + ONE dependent 8-byte read per lane at the head of what the previous launch wrote (no reduction)", vec, ns, part, rec, stream, aux, out);
  run<192>("... the same, handed on through LDS behind a workgroup barrier", vec, ns, part, rec, stream, aux, out);
  run<31, 4>("everything, as 256 workgroups x 1024 threads (four row blocks side by side, 256 partial slots)", vec, ns, part, rec, stream, aux, out);
  run<30, 4>("the same without the scalar fold", vec, ns, part, rec, stream, aux, out);
  run<31, 2>("everything, as 512 workgroups x 512 threads (two row blocks side by side, 512 partial slots)", vec, ns, part, rec, stream, aux, out);
  { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms = 0; const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); for (int k = 1; k <= iters; k++) hipLaunchKernelGGL(k_tall, dim3(G / 2), dim3(512), 0, 0, vec, ns, k, part, rec, stream, aux, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    std::printf("%-100s %6.2f us per launch\n", "everything, TALL blocks: 512 workgroups x 512 threads, 400 rows / 2000 entries, window 400, own 200", 1e3 * ms / iters); }
  return 0;
}
