// microbench.hip -- gfx950 instruction-throughput probes for the Goldilocks arithmetic choices
// documented in DESIGN.md (not part of the product path).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

__device__ __forceinline__ u64 reduce128(u64 lo, u64 hi){
  u32 hh=(u32)(hi>>32), hl=(u32)hi;
  u64 t0=lo-hh; if(lo<hh) t0-=0xFFFFFFFFu;
  u64 t1=((u64)hl<<32)-hl;
  u64 r=t0+t1; if(r<t1) r+=0xFFFFFFFFu;
  return r;
}
__device__ __forceinline__ u64 mulmod(u64 a,u64 b){
  u32 a0=(u32)a,a1=(u32)(a>>32),b0=(u32)b,b1=(u32)(b>>32);
  u64 p00=(u64)a0*b0; u64 p01=(u64)a0*b1+(p00>>32); u64 p10=(u64)a1*b0+(u32)p01;
  u64 p11=(u64)a1*b1+(p01>>32)+(p10>>32);
  return reduce128((p10<<32)|(u32)p00,p11);
}
template<int MODE> __global__ void __launch_bounds__(256) k(u64* x, int iters){
  int i=blockIdx.x*blockDim.x+threadIdx.x;
  u64 a0=x[i],a1=a0^0x1234567,a2=a0+77,a3=a0*3+1; u64 b=a0|1;
  if(MODE==0){ for(int k=0;k<iters;k++){ a0=mulmod(a0,b);a1=mulmod(a1,b);a2=mulmod(a2,b);a3=mulmod(a3,b);} }
  if(MODE==1){ u32 c0=a0,c1=a1,c2=a2,c3=a3,bb=b; for(int k=0;k<iters;k++){ // mad_u64_u32
      a0=(u64)c0*bb+a0; a1=(u64)c1*bb+a1; a2=(u64)c2*bb+a2; a3=(u64)c3*bb+a3; c0=a0>>7; c1=a1>>9; c2=a2>>11; c3=a3>>13;} }
  if(MODE==2){ u32 c0=a0,c1=a1,c2=a2,c3=a3,bb=b; for(int k=0;k<iters;k++){ c0=c0*bb+1; c1=c1*bb+2; c2=c2*bb+3; c3=c3*bb+4;} a0=c0;a1=c1;a2=c2;a3=c3; } // mul_lo
  if(MODE==3){ u32 c0=a0,c1=a1,c2=a2,c3=a3,bb=b; for(int k=0;k<iters;k++){ c0=__umulhi(c0,bb)+c0; c1=__umulhi(c1,bb)+c1; c2=__umulhi(c2,bb)+c2; c3=__umulhi(c3,bb)+c3;} a0=c0;a1=c1;a2=c2;a3=c3; }
  if(MODE==4){ for(int k=0;k<iters;k++){ a0+=b; a1+=a0; a2+=a1; a3+=a2; } } // 64-bit add
  if(MODE==5){ double d0=a0,d1=a1,d2=a2,d3=a3,e=1.0000001; for(int k=0;k<iters;k++){ d0=fma(d0,e,1.0); d1=fma(d1,e,2.0); d2=fma(d2,e,3.0); d3=fma(d3,e,4.0);} a0=d0;a1=d1;a2=d2;a3=d3; }
  if(MODE==6){ u32 c0=a0&0xffffff,c1=a1&0xffffff,c2=a2&0xffffff,c3=a3&0xffffff,bb=b&0xffffff; for(int k=0;k<iters;k++){ c0=__umul24(c0,bb)+1; c1=__umul24(c1,bb)+2; c2=__umul24(c2,bb)+3; c3=__umul24(c3,bb)+4;} a0=c0;a1=c1;a2=c2;a3=c3; }
  x[i]=a0+a1+a2+a3;
}
__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n){
  size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=st) b[i]=a[i];
}
__global__ void read8(const u64* __restrict__ a, u64* out, size_t n){
  size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x; u64 s=0;
  for(;i<n;i+=st) s+=a[i];
  if(s==0x1234567) out[0]=s;
}
int main(){
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr,0));
  printf("device %s CUs %d clock %d kHz mem %.1f GB L2 %d\n",pr.name,pr.multiProcessorCount,pr.clockRate,pr.totalGlobalMem/1e9,pr.l2CacheSize);
  int blocks=256*8, thr=256; u64* x; CK(hipMalloc(&x,(size_t)blocks*thr*8)); CK(hipMemset(x,1,(size_t)blocks*thr*8));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[]={"mulmod(goldilocks)","mad_u64_u32","mul_lo_u32(+add)","mul_hi_u32(+add)","add_u64","fma_f64","mul_u24(+add)"};
  int iters=4096;
  for(int m=0;m<7;m++){
    for(int rep=0;rep<2;rep++){
      CK(hipEventRecord(e0));
      switch(m){case 0:k<0><<<blocks,thr>>>(x,iters);break;case 1:k<1><<<blocks,thr>>>(x,iters);break;case 2:k<2><<<blocks,thr>>>(x,iters);break;
        case 3:k<3><<<blocks,thr>>>(x,iters);break;case 4:k<4><<<blocks,thr>>>(x,iters);break;case 5:k<5><<<blocks,thr>>>(x,iters);break;case 6:k<6><<<blocks,thr>>>(x,iters);break;}
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      if(rep) printf("%-22s %8.3f ms  %8.2f Gop/s (per-lane ops)\n",names[m],ms,(double)blocks*thr*iters*4/ms/1e6);
    }
  }
  size_t bytes=(size_t)4<<30; uint4 *a,*b; CK(hipMalloc(&a,bytes)); CK(hipMalloc(&b,bytes)); CK(hipMemset(a,1,bytes)); CK(hipMemset(b,2,bytes));
  for(int g: {2048,4096,8192,16384}){
    for(int rep=0;rep<3;rep++){ CK(hipEventRecord(e0)); copy16<<<g,256>>>(a,b,bytes/16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      if(rep==2) printf("copy16 grid %5d: %.3f ms  %.1f GB/s (r+w)\n",g,ms,2.0*bytes/ms/1e6);}
    for(int rep=0;rep<3;rep++){ CK(hipEventRecord(e0)); read8<<<g,256>>>((u64*)a,x,bytes/8); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      if(rep==2) printf("read8  grid %5d: %.3f ms  %.1f GB/s (r)\n",g,ms,1.0*bytes/ms/1e6);}
  }
  return 0;
}
