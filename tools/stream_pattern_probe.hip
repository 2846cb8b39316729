// How fast can the (row, seg) 16-byte plane pattern of xc_fast_kernels.hpp stream 4 AO planes
// (nao=114, ngrid=143556) at different workgroup sizes / occupancies / prefetch depths?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
template<int THREADS, int DEPTH>  // DEPTH sub-tiles of loads in flight before consuming
__global__ __launch_bounds__(THREADS) void k_stream(long ngrid, int nao, const double* __restrict__ p0, const double* __restrict__ p1,
                                                    const double* __restrict__ p2, const double* __restrict__ p3, double* out){
  constexpr int ROWS = THREADS/16;
  const int row = threadIdx.x>>4, seg = threadIdx.x&15;
  const long ntile = (ngrid+ROWS-1)/ROWS;
  double s=0;
  double2 v[DEPTH][16];
  long t = blockIdx.x;
  auto issue=[&](int d,long tt){ long g=tt*ROWS+row; bool ok = tt<ntile && g<ngrid; size_t ro=(size_t)(ok?g:0)*nao;
#pragma unroll
    for(int j=0;j<4;j++){ int c=32*j+2*seg; int cc = c<nao-2?c:nao-2;
      v[d][4*j+0]=*(const double2*)(p0+ro+cc); v[d][4*j+1]=*(const double2*)(p1+ro+cc);
      v[d][4*j+2]=*(const double2*)(p2+ro+cc); v[d][4*j+3]=*(const double2*)(p3+ro+cc);} };
#pragma unroll
  for(int d=0;d<DEPTH;d++) issue(d, t+(long)d*gridDim.x);
  for(; t<ntile; t+=(long)DEPTH*gridDim.x){
#pragma unroll
    for(int d=0;d<DEPTH;d++){
#pragma unroll
      for(int i=0;i<16;i++) s+=v[d][i].x+v[d][i].y;
      issue(d, t+(long)(d+DEPTH)*gridDim.x);
    }
  }
  if(s==1.234e-300) out[0]=s;
}
template<int THREADS,int DEPTH> void run(int wgs_per_cu,int ncu,long ngrid,int nao,double* p,double* out){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); size_t plane=(size_t)ngrid*nao;
  int blocks=ncu*wgs_per_cu;
  auto L=[&]{ hipLaunchKernelGGL((k_stream<THREADS,DEPTH>),dim3(blocks),dim3(THREADS),0,0,ngrid,nao,p,p+plane,p+2*plane,p+3*plane,out); };
  L(); hipDeviceSynchronize(); hipEventRecord(e0,0); for(int r=0;r<10;r++) L(); hipEventRecord(e1,0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); ms/=10;
  printf("threads=%d depth=%d WGs/CU=%d: %.1f us  %.0f GB/s\n",THREADS,DEPTH,wgs_per_cu,ms*1e3,4.0*plane*8/ms*1e-6);
}
int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0)); int ncu=prop.multiProcessorCount;
  long ngrid=143556; int nao=114; size_t plane=(size_t)ngrid*nao; double* p; CK(hipMalloc(&p,plane*8*4+4096)); CK(hipMemset(p,0,plane*8*4)); double* out; CK(hipMalloc(&out,64));
  run<256,1>(1,ncu,ngrid,nao,p,out); run<256,2>(1,ncu,ngrid,nao,p,out); run<256,3>(1,ncu,ngrid,nao,p,out); run<128,2>(1,ncu,ngrid,nao,p,out);
  run<512,1>(1,ncu,ngrid,nao,p,out); run<512,2>(1,ncu,ngrid,nao,p,out);
  run<512,1>(2,ncu,ngrid,nao,p,out); run<512,1>(4,ncu,ngrid,nao,p,out);
  run<256,1>(2,ncu,ngrid,nao,p,out); run<256,1>(4,ncu,ngrid,nao,p,out); run<256,2>(4,ncu,ngrid,nao,p,out); run<256,1>(8,ncu,ngrid,nao,p,out);
  return 0;
}
