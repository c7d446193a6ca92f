#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o){ unsigned a=0x01020304u,b=0x04030201u; o[threadIdx.x]=__builtin_amdgcn_sad_u8(a,b,threadIdx.x); }
int main(){ hipDeviceProp_t p; hipGetDeviceProperties(&p,0); printf("%s CUs=%d clk=%d mem=%zu arch=%s\n",p.name,p.multiProcessorCount,p.clockRate,p.totalGlobalMem,p.gcnArchName);
 unsigned* d; hipMalloc(&d,256); k<<<1,64>>>(d); unsigned h[64]; hipMemcpy(h,d,256,hipMemcpyDeviceToHost); printf("sad=%u %u\n",h[0],h[5]); return 0;}
