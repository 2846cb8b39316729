// Diagnostic: where do the cycles of k_vxc_ws go?  (s_memtime shares per role; never quote its run time)
#define QCDFT_STAMPS 1
#include "../quantum_compute_dft_amd/csrc/xc_ws_kernels.hpp"
#include <cstdio>
#include <vector>
#include <random>
using namespace qcdft;
int main(){
  long ngrid=143556; int nao=114; size_t plane=(size_t)ngrid*nao;
  std::vector<double> h(plane*4); std::mt19937_64 rng(1); std::normal_distribution<double> N(0,0.3); for(auto&x:h) x=N(rng);
  double *d,*coef,*slabs; hipMalloc(&d,plane*32); hipMemcpy(d,h.data(),plane*32,hipMemcpyHostToDevice);
  hipMalloc(&coef,ngrid*32); hipMemcpy(coef,h.data(),ngrid*32,hipMemcpyHostToDevice); hipMalloc(&slabs,(size_t)256*nao*nao*8);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int rep=0;rep<3;rep++){
    hipEventRecord(e0,0);
    hipLaunchKernelGGL((k_vxc_ws<8,true,true,false>),dim3(256),dim3(512),0,0,ngrid,nao,d,d+plane,d+2*plane,d+3*plane,coef,slabs,0);
    hipEventRecord(e1,0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
    std::vector<unsigned long long> st(256*8*4); hipMemcpyFromSymbol(st.data(),HIP_SYMBOL(g_stamps),st.size()*8);
    double m_work=0,m_bar=0,l_wait=0,l_work=0,l_bar=0,l_q=0;
    for(int b=0;b<256;b++) for(int w=0;w<8;w++){ auto*p=&st[(b*8+w)*4]; if(w<4){m_work+=p[0]; m_bar+=p[1];} else {l_wait+=p[0]; l_work+=p[1]; l_bar+=p[2]; l_q+=p[3];} }
    if(rep==2){ for(int b: {0,100}) { printf("block %d per wave [w: s0 s1 s2 s3]:",b); for(int w=0;w<8;w++){auto*p=&st[(b*8+w)*4]; printf("  w%d: %llu %llu %llu %llu",w,p[0],p[1],p[2],p[3]);} printf("\n"); } }
    double n=256*4*37.05; // per wave and per step (35.05 sub-tiles per workgroup + 2 ring steps)
    printf("kernel %.1f us | MFMA waves: work %.0f  barrier-wait %.0f cycles | loader waves: load-wait %.0f  stage+issue %.0f  barrier-wait %.0f | of stage+issue: Q math+LDS writes %.0f (cycles per wave per STEP)\n",
       ms*1e3,m_work/n,m_bar/n,l_wait/n,l_work/n,l_bar/n,l_q/n);
  }
  return 0;
}
