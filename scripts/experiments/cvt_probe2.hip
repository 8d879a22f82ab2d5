#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* in, const float* sc, unsigned* out) {
  unsigned w = in[threadIdx.x]; float s = sc[threadIdx.x];
  bf16x2 a0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 0);
  bf16x2 a1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 1);
  bf16x2 a2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 2);
  bf16x2 a3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 3);
  bf16x2 c0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, s, false);
  bf16x2 c1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, s, true);
  bf16x2 d0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(w, s, false);
  unsigned* o = out + threadIdx.x*8;
  o[0]=__builtin_bit_cast(unsigned,a0); o[1]=__builtin_bit_cast(unsigned,a1); o[2]=__builtin_bit_cast(unsigned,a2); o[3]=__builtin_bit_cast(unsigned,a3);
  o[4]=__builtin_bit_cast(unsigned,c0); o[5]=__builtin_bit_cast(unsigned,c1); o[6]=__builtin_bit_cast(unsigned,d0);
}
static float bf(unsigned h){ unsigned u=h<<16; float f; f=__builtin_bit_cast(float,u); return f; }
int main(){
  const int N=8; unsigned hin[N]={0x76543210u,0xFEDCBA98u,0x76543210u,0x7E7F0138u,0x76543210u,0x76543210u,0x76543210u,0x76543210u};
  float hs[N]={-1.0f,-3.5f,0.25f,1.0f,__builtin_bit_cast(float,0x00000000u),__builtin_bit_cast(float,0x00400000u), __builtin_bit_cast(float,0x7F800000u), __builtin_bit_cast(float,0x7F000000u)};
  unsigned *din,*dout; float* ds; hipMalloc(&din,N*4); hipMalloc(&ds,N*4); hipMalloc(&dout,N*32);
  hipMemcpy(din,hin,N*4,hipMemcpyHostToDevice); hipMemcpy(ds,hs,N*4,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k,dim3(1),dim3(N),0,0,din,ds,dout); unsigned ho[N*8]; hipMemcpy(ho,dout,N*32,hipMemcpyDeviceToHost);
  for(int i=0;i<N;i++){ printf("in=%08x scale=%g\n fp4:",hin[i],hs[i]); for(int j=0;j<4;j++) printf(" [%g %g]",bf(ho[i*8+j]&0xffff),bf(ho[i*8+j]>>16));
    printf("\n fp8:"); for(int j=4;j<6;j++) printf(" [%g %g]",bf(ho[i*8+j]&0xffff),bf(ho[i*8+j]>>16)); printf("\n bf8: [%g %g]\n",bf(ho[i*8+6]&0xffff),bf(ho[i*8+6]>>16)); }
  return 0; }
