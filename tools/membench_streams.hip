#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
// MODE 0: SoA (8 columns) read + write ; 1: AoS per-lane 64B (4 x double2) ; 2: AoS through LDS transpose (coalesced 16B/lane)
// WMODE 3: SoA with non-temporal stores (the variant the fused kernel ships with)
template<int RMODE, int WMODE>
__global__ __launch_bounds__(256) void k_like(const double2* __restrict__ P, const double* __restrict__ I0, const double* __restrict__ J0,
   double* __restrict__ It, double* __restrict__ G, double* __restrict__ J, unsigned N, int rows) {
  __shared__ double tile[4][64*8+8];
  const int t = blockIdx.y; const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double2* p = P + (size_t)t*N; const double* i0 = I0 + (size_t)t*N; const double* j0 = J0 + (size_t)t*N*8;
  double* it = It + (size_t)t*N; double* g = G + (size_t)t*N*2; double* j = J + (size_t)t*N*8;
  unsigned base = blockIdx.x*(256u*rows) + threadIdx.x;
  double acc = 0;
  #pragma unroll 1
  for (int k=0;k<rows;++k) {
    unsigned i = base + k*256u;
    if (i >= N) break;
    double2 cp = p[i]; double ci = i0[i]; double cj[8];
    if (RMODE == 3) { for (int s=0;s<8;++s) cj[s] = ci * s; }
    else if (RMODE == 0) { for (int s=0;s<8;++s) cj[s] = j0[(unsigned)s*N+i]; }
    else if (RMODE == 1) { const double2* q = reinterpret_cast<const double2*>(j0 + (size_t)i*8); for (int s=0;s<4;++s){ double2 v=q[s]; cj[2*s]=v.x; cj[2*s+1]=v.y; } }
    else { // coalesced: wave tile of 64 px * 64 B = 4 KB ; lane loads 16B chunks c = lane + 64*m
      unsigned wbase = (i - lane); const double2* q = reinterpret_cast<const double2*>(j0 + (size_t)wbase*8);
      for (int m=0;m<4;++m){ double2 v = q[lane + 64*m]; int e = (lane + 64*m)*2; tile[wv][e] = v.x; tile[wv][e+1] = v.y; }
      __builtin_amdgcn_wave_barrier();
      for (int s=0;s<8;++s) cj[s] = tile[wv][lane*8+s];
      __builtin_amdgcn_wave_barrier();
    }
    double v = cp.x + cp.y + ci;
    acc += v;
    if (WMODE == 4) { for (int s=0;s<8;++s) acc += cj[s]; continue; }
    if (WMODE == 3) { __builtin_nontemporal_store(v, &it[i]); __builtin_nontemporal_store(v*2, &g[i]); __builtin_nontemporal_store(v*3, &g[N+i]);
      for (int s=0;s<8;++s) __builtin_nontemporal_store(cj[s] + v, &j[(unsigned)s*N+i]); }
    else { it[i] = v; g[i] = v*2; g[N+i] = v*3; }
    if (WMODE == 3) {}
    else if (WMODE == 0) { for (int s=0;s<8;++s) j[(unsigned)s*N+i] = cj[s] + v; }
    else if (WMODE == 1) { double2* q = reinterpret_cast<double2*>(j + (size_t)i*8); for (int s=0;s<4;++s) q[s] = make_double2(cj[2*s]+v, cj[2*s+1]+v); }
    else {
      for (int s=0;s<8;++s) tile[wv][lane*8+s] = cj[s] + v;
      __builtin_amdgcn_wave_barrier();
      unsigned wbase = (i - lane); double2* q = reinterpret_cast<double2*>(j + (size_t)wbase*8);
      for (int m=0;m<4;++m){ int e = (lane + 64*m)*2; q[lane + 64*m] = make_double2(tile[wv][e], tile[wv][e+1]); }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (acc == 1.2345) it[0] = acc;
}
// a small latency-bound kernel (64 workgroups of one wave, ~8 us) like k_finish_track, to interleave between the big launches
__global__ __launch_bounds__(64) void k_small(double *x, int iters) {
  double v = x[blockIdx.x * 64 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0000001 + x[(blockIdx.x * 64 + ((threadIdx.x + i) & 63))] * 1e-9;
  x[blockIdx.x * 64 + threadIdx.x] = v;
}
__global__ void k_fill(double *p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = 1.0 + (double)h * 1e-9;
  }
}
int main(int argc, char **argv){
  const unsigned N=40000+ (64 - 40000%64)%64; const int T = argc > 1 ? atoi(argv[1]) : 64;
  const int dyn_lds = argc > 2 ? atoi(argv[2]) : 0;
  const int interleave = argc > 3 ? atoi(argv[3]) : 0;   // >0: run k_small(interleave iterations) between the timed launches
  double *X; CK(hipMalloc(&X, 64*64*8)); CK(hipMemset(X, 0, 64*64*8));   // bytes of dynamic LDS per workgroup: caps the resident workgroups per CU (160 KB LDS)   // N multiple of 64 for the tile variants
  double2* P; double *I0,*J0,*It,*G,*J;
  CK(hipMalloc(&P,(size_t)N*T*16)); CK(hipMalloc(&I0,(size_t)N*T*8)); CK(hipMalloc(&J0,(size_t)N*T*64));
  CK(hipMalloc(&It,(size_t)N*T*8)); CK(hipMalloc(&G,(size_t)N*T*16)); CK(hipMalloc(&J,(size_t)N*T*64));
  CK(hipMemset(P,0,(size_t)N*T*16)); CK(hipMemset(I0,0,(size_t)N*T*8)); CK(hipMemset(J0,0,(size_t)N*T*64));
  if (argc > 4 && atoi(argv[4])) {   // non-zero payload (the default buffers are all zeros)
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double*)P, (size_t)N*T*2, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, I0, (size_t)N*T, 2u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, J0, (size_t)N*T*8, 3u);
    CK(hipDeviceSynchronize());
  }
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto run=[&](const char* name, auto launch, double bytes){
    for(int i=0;i<3;++i) launch();
    float ms = 0;
    if (!interleave) { hipEventRecord(a); for(int i=0;i<20;++i) launch(); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms,a,b); ms/=20; }
    else {
      for (int i=0;i<20;++i) {
        hipLaunchKernelGGL(k_small, dim3(64), dim3(64), 0, 0, X, interleave);
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t,a,b); ms += t;
      }
      ms /= 20;
    }
    printf("%-40s %8.2f us  %7.1f GB/s\n", name, ms*1e3, bytes/ms/1e6);
  };
  double bytes = (double)N*T*176;
  int total_rows = (N+255)/256;
  for (int rows : {4, 20}) {
    int nb = (total_rows + rows-1)/rows; char nm[96];
#define RUN(R,W) snprintf(nm,96,"read%d write%d rows%d", R, W, rows); run(nm,[&]{hipLaunchKernelGGL((k_like<R,W>),dim3(nb,T),dim3(256),dyn_lds,0,P,I0,J0,It,G,J,N,rows);}, bytes);
    RUN(0,0) RUN(0,3) RUN(3,3) RUN(0,4) RUN(3,0) RUN(0,1) RUN(0,2) RUN(1,1) RUN(2,2) RUN(1,0) RUN(2,0)
  }
  return 0;
}
