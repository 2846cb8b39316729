// Does a VALU-only wave get starved by a back-to-back fp64-MFMA wave on the same SIMD?
// Block = 512 threads: waves 0-3 MFMA loop, waves 4-7 VALU loop (one of each per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
template<int MODE> // VALU wave flavour: 0 = v_add_u32 chain x8 indep, 1 = v_fma_f64 x8 indep, 2 = ds_write_b128
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* clk, int mf_iters, int va_iters, int mfma_on, int prio){
  __shared__ double lds[4096];
  const int wave=__builtin_amdgcn_readfirstlane(threadIdx.x>>6);
  unsigned long long c0=__builtin_amdgcn_s_memtime();
  if(wave<4){
    d4 a0={0,0,0,0},a1=a0,a2=a0,a3=a0; double x=threadIdx.x*1e-3, y=1.0001;
    if(mfma_on) for(int i=0;i<mf_iters;i++){ a0=__builtin_amdgcn_mfma_f64_16x16x4f64(x,y,a0,0,0,0); a1=__builtin_amdgcn_mfma_f64_16x16x4f64(x,y,a1,0,0,0);
      a2=__builtin_amdgcn_mfma_f64_16x16x4f64(x,y,a2,0,0,0); a3=__builtin_amdgcn_mfma_f64_16x16x4f64(x,y,a3,0,0,0);}
    out[blockIdx.x*512+threadIdx.x]=a0[0]+a1[1]+a2[2]+a3[3];
  } else {
    if(prio) __builtin_amdgcn_s_setprio(3);
    if(MODE==0){ unsigned v[8]; for(int i=0;i<8;i++) v[i]=threadIdx.x+i;
      for(int it=0;it<va_iters;it++){
#pragma unroll
        for(int i=0;i<8;i++) v[i]=v[i]*3u+7u; }
      unsigned s=0; for(int i=0;i<8;i++) s+=v[i]; out[blockIdx.x*512+threadIdx.x]=s; }
    if(MODE==1){ double v[8]; for(int i=0;i<8;i++) v[i]=threadIdx.x+i;
      for(int it=0;it<va_iters;it++){
#pragma unroll
        for(int i=0;i<8;i++) v[i]=__builtin_fma(v[i],1.0000001,1e-9); }
      double s=0; for(int i=0;i<8;i++) s+=v[i]; out[blockIdx.x*512+threadIdx.x]=s; }
    if(MODE==2){ double2 v=make_double2(threadIdx.x,1.0); int o=(threadIdx.x-256)*2;
      for(int it=0;it<va_iters;it++){
#pragma unroll
        for(int i=0;i<8;i++) *reinterpret_cast<double2*>(&lds[(o+i*512)&4094])=v; }
      out[blockIdx.x*512+threadIdx.x]=lds[threadIdx.x]; }
  }
  unsigned long long c1=__builtin_amdgcn_s_memtime();
  if((threadIdx.x&63)==0) clk[blockIdx.x*8+wave]=c1-c0;
}
template<int MODE> void run(const char* name,int mfma_on,int prio,double* out,unsigned long long* clk){
  int mf=2000, va=4000;
  hipLaunchKernelGGL((k<MODE>),dim3(256),dim3(512),0,0,out,clk,mf,va,mfma_on,prio); hipDeviceSynchronize();
  hipLaunchKernelGGL((k<MODE>),dim3(256),dim3(512),0,0,out,clk,mf,va,mfma_on,prio); hipDeviceSynchronize();
  std::vector<unsigned long long> h(256*8); hipMemcpy(h.data(),clk,h.size()*8,hipMemcpyDeviceToHost);
  double m=0,v=0; for(int b=0;b<256;b++) for(int w=0;w<8;w++){ if(w<4) m+=h[b*8+w]; else v+=h[b*8+w]; } m/=1024; v/=1024;
  printf("%-14s mfma_on=%d prio=%d : MFMA wave %.1f cyc/MFMA ; VALU wave %.1f cyc/instr (8 per iter)\n",name,mfma_on,prio,m/(4.0*mf),v/(8.0*va));
}
int main(){ double* out; hipMalloc(&out,256*512*8); unsigned long long* clk; hipMalloc(&clk,256*8*8);
  run<0>("v_mad_u32",0,0,out,clk); run<0>("v_mad_u32",1,0,out,clk); run<0>("v_mad_u32",1,1,out,clk);
  run<1>("v_fma_f64",0,0,out,clk); run<1>("v_fma_f64",1,0,out,clk); run<1>("v_fma_f64",1,1,out,clk);
  run<2>("ds_write_b128",0,0,out,clk); run<2>("ds_write_b128",1,0,out,clk); run<2>("ds_write_b128",1,1,out,clk);
  return 0; }
