#include "libm_probe_host.h"
#include <stdio.h>
#include <omp.h>
#include <vector>
int main(){
  struct T{const char*name; uint32_t lo,hi; int fn;};
  // float bit ranges: [0,1] -> 0..0x3f800000 ; [0, 2pi] -> up to 0x40c90fdb ; negatives for sin/cos not needed
  T tests[]={{"acosf [0,1]",0,0x3f800000u,1},{"acosf [-1,0)",0x80000000u,0xbf800000u,1},{"sinf [0,7]",0,0x40e00000u,2},{"cosf [0,7]",0,0x40e00000u,3},
             {"powf(x,1/2.2f) [0,2]",0,0x40000000u,0},{"powf(x,1/2.2f) (2,inf]",0x40000001u,0x7f800000u,0},{"powf(x,2.2f) [0,1]",0,0x3f800000u,4}};
  const float ig=1.f/2.2f;
  for(auto&t:tests){
    long long bad=0; uint32_t firstbad=0; 
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for(long long u=t.lo; u<=(long long)t.hi; ++u){
      float x=asfloat((uint32_t)u), a,b;
      switch(t.fn){case 0:a=powf(x,ig);b=pt_powf_host(x,ig);break;case 1:a=acosf(x);b=pt_acosf_host(x);break;case 2:a=sinf(x);b=pt_sinf_host(x);break;case 3:a=cosf(x);b=pt_cosf_host(x);break;default:a=powf(x,2.2f);b=pt_powf_host(x,2.2f);}
      if(asuint(a)!=asuint(b) && !(a!=a&&b!=b)){ bad++; if(!firstbad){
        #pragma omp critical
        if(!firstbad){firstbad=(uint32_t)u; }}}
    }
    printf("%-26s mismatches %lld of %lld", t.name, bad, (long long)t.hi-t.lo+1);
    if(bad){ float x=asfloat(firstbad); printf("  e.g. x=%a", x);} printf("\n"); fflush(stdout);
  }
}
