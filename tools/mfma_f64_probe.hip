// Probe for facts the local guides do not state: fp64 MFMA operand/result
// layout, its sustained rate (vs v_fma_f64), and streaming HBM bandwidth.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

__global__ void k_layout(const double* A, const double* B, double* D){ // A 16x4 row-major, B 4x16 row-major
  int l=threadIdx.x; d4 acc={0,0,0,0};
  acc=__builtin_amdgcn_mfma_f64_16x16x4f64(A[(l&15)*4+(l>>4)], B[(l>>4)*16+(l&15)], acc,0,0,0);
  for(int r=0;r<4;r++) D[((l>>4)+4*r)*16+(l&15)]=acc[r];
}
template<int NACC>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters, double a, double b){
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double x=a+threadIdx.x*1e-9, y=b;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(x,y,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ __launch_bounds__(256) void k_fma(double* out,int iters,double a,double b){
  double v[8]; for(int i=0;i<8;i++) v[i]=threadIdx.x*1e-9+i;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<8;i++) v[i]=__builtin_fma(v[i],a,b);
  }
  double s=0; for(int i=0;i<8;i++) s+=v[i]; out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ __launch_bounds__(256) void k_read(const double2* p, size_t n, double* out){
  double s=0; size_t stride=(size_t)gridDim.x*blockDim.x;
  for(size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=stride){ double2 v=p[i]; s+=v.x+v.y; }
  if(s==1.2345e-300) out[0]=s;
}
__global__ __launch_bounds__(256) void k_write(double2* p, size_t n){
  size_t stride=(size_t)gridDim.x*blockDim.x;
  for(size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=stride) p[i]=make_double2(1.0,2.0);
}
int main(){
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop,0));
  printf("device %s CUs %d clock %d kHz\n",prop.gcnArchName,prop.multiProcessorCount,prop.clockRate);
  // layout
  std::vector<double> A(64),B(64),D(256),R(256,0.0);
  for(int i=0;i<64;i++){A[i]=(i*7)%11-5; B[i]=(i*5)%13-6;}
  for(int i=0;i<16;i++)for(int j=0;j<16;j++){double s=0;for(int k=0;k<4;k++)s+=A[i*4+k]*B[k*16+j];R[i*16+j]=s;}
  double *dA,*dB,*dD; CK(hipMalloc(&dA,512));CK(hipMalloc(&dB,512));CK(hipMalloc(&dD,2048));
  CK(hipMemcpy(dA,A.data(),512,hipMemcpyHostToDevice));CK(hipMemcpy(dB,B.data(),512,hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_layout,dim3(1),dim3(64),0,0,dA,dB,dD); CK(hipMemcpy(D.data(),dD,2048,hipMemcpyDeviceToHost));
  int bad=0; for(int i=0;i<256;i++) if(D[i]!=R[i]) bad++;
  printf("f64 MFMA layout (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15]): %s (%d mismatches)\n",bad?"WRONG":"OK",bad);
  // rates
  double* out; CK(hipMalloc(&out,sizeof(double)*256*4096));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time=[&](auto launch){ launch(); hipDeviceSynchronize(); hipEventRecord(e0,0); for(int r=0;r<5;r++) launch(); hipEventRecord(e1,0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); return ms/5; };
  int ncu=prop.multiProcessorCount; int iters=20000;
  for(int wps=1; wps<=2; wps++){
    int blocks=ncu*wps;
    float ms=time([&]{hipLaunchKernelGGL(k_rate<4>,dim3(blocks),dim3(256),0,0,out,iters,1.0001,0.9999);});
    double fl=(double)blocks*4*iters*4*2048.0; printf("MFMA f64 16x16x4, 4 acc, %d wave/SIMD: %.1f TFLOP/s (%.3f ms)\n",wps,fl/ms*1e-9,ms);
    ms=time([&]{hipLaunchKernelGGL(k_rate<1>,dim3(blocks),dim3(256),0,0,out,iters,1.0001,0.9999);});
    fl=(double)blocks*4*iters*1*2048.0; printf("MFMA f64 16x16x4, 1 acc (dependent), %d wave/SIMD: %.1f TFLOP/s\n",wps,fl/ms*1e-9);
  }
  for(int wps=1; wps<=4; wps*=2){
    int blocks=ncu*wps;
    float ms=time([&]{hipLaunchKernelGGL(k_fma,dim3(blocks),dim3(256),0,0,out,iters,1.0000001,1e-9);});
    double fl=(double)blocks*256*iters*8*2.0; printf("v_fma_f64, %d wave/SIMD: %.1f TFLOP/s\n",wps,fl/ms*1e-9);
  }
  // HBM
  size_t bytes=(size_t)2<<30; double2* buf; CK(hipMalloc(&buf,bytes)); size_t n=bytes/16;
  float ms=time([&]{hipLaunchKernelGGL(k_write,dim3(ncu*8),dim3(256),0,0,buf,n);}); printf("HBM write 2 GiB: %.0f GB/s\n",bytes/ms*1e-6);
  ms=time([&]{hipLaunchKernelGGL(k_read,dim3(ncu*8),dim3(256),0,0,buf,n,out);}); printf("HBM read  2 GiB: %.0f GB/s\n",bytes/ms*1e-6);
  return 0;
}
