// Probe 2: cycles per fp64 MFMA, shader clock under load, waves/SIMD and accumulator sweep,
// MFMA fed from LDS, MFMA + VALU co-issue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
__device__ inline double rnd(unsigned i){ i=i*2654435761u+12345u; i^=i>>13; i*=1274126177u; i^=i>>16; return (double)(i&0xFFFFF)/524288.0-1.0; }

template<int NACC,int MODE>  // MODE 0: regs, 1: operands re-read from LDS each MFMA, 2: regs + 2 v_fma_f64 per MFMA
__global__ __launch_bounds__(256) void k_rate(double* out, unsigned long long* clk, int iters){
  __shared__ double lds[2048];
  for(int i=threadIdx.x;i<2048;i+=256) lds[i]=rnd(i+7*blockIdx.x);
  __syncthreads();
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  const int lane=threadIdx.x&63;
  double x=rnd(threadIdx.x+blockIdx.x*977), y=rnd(threadIdx.x*31+5);
  double v0=x,v1=y;
  unsigned long long c0=__builtin_amdgcn_s_memtime(), r0=__builtin_amdgcn_s_memrealtime();
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<NACC;i++){
      if(MODE==1){ x=lds[(it*NACC+i)%28*64+lane]; y=lds[((it*NACC+i)%28+2)*64+lane]; }
      acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(x,y,acc[i],0,0,0);
      if(MODE==2){ v0=__builtin_fma(v0,1.0000001,1e-9); v1=__builtin_fma(v1,0.9999999,1e-9); }
    }
  }
  unsigned long long c1=__builtin_amdgcn_s_memtime(), r1=__builtin_amdgcn_s_memrealtime();
  double s=v0+v1;
#pragma unroll
  for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if(threadIdx.x==0){ clk[2*blockIdx.x]=c1-c0; clk[2*blockIdx.x+1]=r1-r0; }
}
template<int NACC,int MODE>
int run(const char* tag,int ncu,int wps,int iters,double* out,unsigned long long* clk){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks=ncu*wps;
  hipLaunchKernelGGL((k_rate<NACC,MODE>),dim3(blocks),dim3(256),0,0,out,clk,iters); hipDeviceSynchronize();
  hipEventRecord(e0,0);
  for(int r=0;r<3;r++) hipLaunchKernelGGL((k_rate<NACC,MODE>),dim3(blocks),dim3(256),0,0,out,clk,iters);
  hipEventRecord(e1,0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=3;
  std::vector<unsigned long long> h(2*blocks); hipMemcpy(h.data(),clk,16*blocks,hipMemcpyDeviceToHost);
  double cyc=0,rt=0; for(int b=0;b<blocks;b++){cyc+=h[2*b]; rt+=h[2*b+1];} cyc/=blocks; rt/=blocks;
  double ghz=cyc/(rt*10.0); // realtime ticks are 10 ns
  double fl=(double)blocks*4*iters*NACC*2048.0;
  printf("%-28s acc=%d waves/SIMD=%d: %6.1f TFLOP/s  %.3f ms  clock %.2f GHz  cycles/MFMA/wave %.1f  (per SIMD %.1f)\n",
         tag,NACC,wps,fl/ms*1e-9,ms,ghz,cyc/((double)iters*NACC),cyc/((double)iters*NACC*wps));
  return 0;
}
int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0)); int ncu=prop.multiProcessorCount;
  double* out; CK(hipMalloc(&out,8*256*ncu*8)); unsigned long long* clk; CK(hipMalloc(&clk,16*ncu*8));
  int it=8000;
  run<1,0>("regs",ncu,1,it*4,out,clk); run<2,0>("regs",ncu,1,it*2,out,clk); run<4,0>("regs",ncu,1,it,out,clk); run<8,0>("regs",ncu,1,it/2,out,clk);
  run<4,0>("regs",ncu,2,it,out,clk); run<8,0>("regs",ncu,2,it/2,out,clk);
  run<4,0>("regs",ncu,4,it,out,clk); run<8,0>("regs",ncu,4,it/2,out,clk); run<2,0>("regs",ncu,8,it*2,out,clk);
  run<4,1>("operands from LDS",ncu,1,it,out,clk); run<4,1>("operands from LDS",ncu,2,it,out,clk); run<4,1>("operands from LDS",ncu,4,it,out,clk);
  run<4,2>("regs + 2 v_fma_f64/MFMA",ncu,1,it,out,clk); run<4,2>("regs + 2 v_fma_f64/MFMA",ncu,2,it,out,clk);
  return 0;
}
