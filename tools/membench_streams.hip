#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
// MODE 0: SoA (8 columns) read + write ; 1: AoS per-lane 64B (4 x double2) ; 2: AoS through LDS transpose (coalesced 16B/lane)
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
    if (RMODE == 0) { for (int s=0;s<8;++s) cj[s] = j0[(unsigned)s*N+i]; }
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
    it[i] = v; g[i] = v*2; g[N+i] = v*3;
    if (WMODE == 0) { for (int s=0;s<8;++s) j[(unsigned)s*N+i] = cj[s] + v; }
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
int main(){
  const unsigned N=40000+ (64 - 40000%64)%64; const int T=64;   // N multiple of 64 for the tile variants
  double2* P; double *I0,*J0,*It,*G,*J;
  CK(hipMalloc(&P,(size_t)N*T*16)); CK(hipMalloc(&I0,(size_t)N*T*8)); CK(hipMalloc(&J0,(size_t)N*T*64));
  CK(hipMalloc(&It,(size_t)N*T*8)); CK(hipMalloc(&G,(size_t)N*T*16)); CK(hipMalloc(&J,(size_t)N*T*64));
  CK(hipMemset(P,0,(size_t)N*T*16)); CK(hipMemset(I0,0,(size_t)N*T*8)); CK(hipMemset(J0,0,(size_t)N*T*64));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  auto run=[&](const char* name, auto launch, double bytes){
    for(int i=0;i<3;++i) launch();
    hipEventRecord(a); for(int i=0;i<20;++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms,a,b); ms/=20;
    printf("%-40s %8.2f us  %7.1f GB/s\n", name, ms*1e3, bytes/ms/1e6);
  };
  double bytes = (double)N*T*176;
  int total_rows = (N+255)/256;
  for (int rows : {4, 20}) {
    int nb = (total_rows + rows-1)/rows; char nm[96];
#define RUN(R,W) snprintf(nm,96,"read%d write%d rows%d", R, W, rows); run(nm,[&]{hipLaunchKernelGGL((k_like<R,W>),dim3(nb,T),dim3(256),0,0,P,I0,J0,It,G,J,N,rows);}, bytes);
    RUN(0,0) RUN(0,1) RUN(0,2) RUN(1,1) RUN(2,2) RUN(1,0) RUN(2,0)
  }
  return 0;
}
