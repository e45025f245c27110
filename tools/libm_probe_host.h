// Candidate restatement of glibc 2.35 float libm routines (shared by the CPU probe).
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#ifndef PT_FMA
#define PT_FMA(a,b,c) ((a)*(b)+(c))
#endif
static inline uint32_t asuint(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float asfloat(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint64_t asuint64(double f){uint64_t u;memcpy(&u,&f,8);return u;}
static inline double asdouble(uint64_t u){double f;memcpy(&f,&u,8);return f;}

// ---------------- sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c)
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4, s1,s2,s3; } sincos_t;
static const sincos_t sincosf_table[2] = {
 { {1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
   -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 },
 { {1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
   0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 } };
static inline uint32_t abstop12(float x){ return (asuint(x)>>20)&0x7ff; }
static inline float sinf_poly(double x,double x2,const sincos_t*p,int n){
  double x3,x4,x6,x7,s,c,c1,c2,s1;
  if((n&1)==0){ x3=x*x2; s1=p->s2+x2*p->s3; x7=x3*x2; s=x+x3*p->s1; return (float)(s+x7*s1);}
  else { x4=x2*x2; c2=p->c3+x2*p->c4; c1=p->c0+x2*p->c1; x6=x4*x2; c=c1+x4*p->c2; return (float)(c+x6*c2);} }
static inline double reduce_fast(double x,const sincos_t*p,int*np){
  double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return x-n*p->hpi; }
static inline float pt_sinf_host(float y){
  double x=y,s; int n; const sincos_t*p=&sincosf_table[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){ s=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return y; return sinf_poly(x,s,p,0);}
  else if(abstop12(y)<abstop12(120.0f)){ x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&sincosf_table[1]; return sinf_poly(x*s,x*x,p,n);}
  return NAN; }
static inline float pt_cosf_host(float y){
  double x=y,s; int n; const sincos_t*p=&sincosf_table[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){ double x2=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return 1.0f; return sinf_poly(x,x2,p,1);}
  else if(abstop12(y)<abstop12(120.0f)){ x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&sincosf_table[1]; return sinf_poly(x*s,x*x,p,n^1);}
  return NAN; }

// ---------------- acosf (sysdeps/ieee754/flt-32/e_acosf.c, fdlibm)
static inline float pt_acosf_host(float x){
  const float one=1.0f, pi=3.1415925026e+00f, pio2_hi=1.5707962513e+00f, pio2_lo=7.5497894159e-08f,
   pS0=1.6666667163e-01f,pS1=-3.2556581497e-01f,pS2=2.0121252537e-01f,pS3=-4.0055535734e-02f,pS4=7.9153501429e-04f,pS5=3.4793309169e-05f,
   qS1=-2.4033949375e+00f,qS2=2.0209457874e+00f,qS3=-6.8828397989e-01f,qS4=7.7038154006e-02f;
  float z,p,q,r,w,s,c,df; int32_t hx=(int32_t)asuint(x), ix=hx&0x7fffffff;
  if(ix==0x3f800000){ if(hx>0) return 0.0f; else return pi+2.0f*pio2_lo; }
  else if(ix>0x3f800000) return (x-x)/(x-x);
  if(ix<0x3f000000){ if(ix<=0x23000000) return pio2_hi+pio2_lo; z=x*x;
    p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; return pio2_hi-(x-(pio2_lo-x*r)); }
  else if(hx<0){ z=(one+x)*0.5f; p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); s=sqrtf(z); r=p/q; w=r*s-pio2_lo; return pi-2.0f*(s+w); }
  else { int32_t idf; z=(one-x)*0.5f; s=sqrtf(z); df=s; idf=(int32_t)asuint(df); df=asfloat((uint32_t)idf&0xfffff000u); c=(z-df*df)/(s+df);
    p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; w=r*s+c; return 2.0f*(df+w); } }

// ---------------- powf (sysdeps/ieee754/flt-32/e_powf.c, e_powf_log2_data.c, e_exp2f_data.c)
static const double powf_log2_tab[16][2] = {
 {0x1.661ec79f8f3bep+0,-0x1.efec65b963019p-2},{0x1.571ed4aaf883dp+0,-0x1.b0b6832d4fca4p-2},{0x1.49539f0f010bp+0,-0x1.7418b0a1fb77bp-2},
 {0x1.3c995b0b80385p+0,-0x1.39de91a6dcf7bp-2},{0x1.30d190c8864a5p+0,-0x1.01d9bf3f2b631p-2},{0x1.25e227b0b8eap+0,-0x1.97c1d1b3b7afp-3},
 {0x1.1bb4a4a1a343fp+0,-0x1.2f9e393af3c9fp-3},{0x1.12358f08ae5bap+0,-0x1.960cbbf788d5cp-4},{0x1.0953f419900a7p+0,-0x1.a6f9db6475fcep-5},
 {0x1p+0,0x0p+0},{0x1.e608cfd9a47acp-1,0x1.338ca9f24f53dp-4},{0x1.ca4b31f026aap-1,0x1.476a9543891bap-3},{0x1.b2036576afce6p-1,0x1.e840b4ac4e4d2p-3},
 {0x1.9c2d163a1aa2dp-1,0x1.40645f0c6651cp-2},{0x1.886e6037841edp-1,0x1.88e9c2c1b9ff8p-2},{0x1.767dcf5534862p-1,0x1.ce0a44eb17bccp-2}};
static const double powf_log2_poly[5]={0x1.27616c9496e0bp-2,-0x1.71969a075c67ap-2,0x1.ec70a6ca7baddp-2,-0x1.7154748bef6c8p-1,0x1.71547652ab82bp0};
static const uint64_t exp2f_tab[32]={
0x3ff0000000000000,0x3fefd9b0d3158574,0x3fefb5586cf9890f,0x3fef9301d0125b51,0x3fef72b83c7d517b,0x3fef54873168b9aa,0x3fef387a6e756238,0x3fef1e9df51fdee1,
0x3fef06fe0a31b715,0x3feef1a7373aa9cb,0x3feedea64c123422,0x3feece086061892d,0x3feebfdad5362a27,0x3feeb42b569d4f82,0x3feeab07dd485429,0x3feea47eb03a5585,
0x3feea09e667f3bcd,0x3fee9f75e8ec5f74,0x3feea11473eb0187,0x3feea589994cce13,0x3feeace5422aa0db,0x3feeb737b0cdc5e5,0x3feec49182a3f090,0x3feed503b23e255d,
0x3feee89f995ad3ad,0x3feeff76f2fb5e47,0x3fef199bdd85529c,0x3fef3720dcef9069,0x3fef5818dcfba487,0x3fef7c97337b9b5f,0x3fefa4afa2a490da,0x3fefd0765b6e4540};
static const double exp2f_poly[3]={0x1.c6af84b912394p-5,0x1.ebfce50fac4f3p-3,0x1.62e42ff0c52d6p-1};
static inline double powf_log2_inline(uint32_t ix){
  uint32_t tmp=ix-0x3f330000u; int i=(tmp>>(23-4))%16; uint32_t top=tmp&0xff800000u; uint32_t iz=ix-top; int k=(int32_t)top>>23;
  double invc=powf_log2_tab[i][0], logc=powf_log2_tab[i][1], z=(double)asfloat(iz);
  double r=z*invc-1, y0=logc+(double)k; const double*A=powf_log2_poly;
  double r2=r*r, y=A[0]*r+A[1], p=A[2]*r+A[3], r4=r2*r2, q=A[4]*r+y0; q=p*r2+q; y=y*r4+q; return y; }
static inline float powf_exp2_inline(double xd,uint32_t sign_bias){
  const double SHIFT=0x1.8p+52/32; double kd=xd+SHIFT; uint64_t ki=asuint64(kd); kd-=SHIFT; double r=xd-kd;
  uint64_t t=exp2f_tab[ki%32]; uint64_t ski=ki+sign_bias; t+=ski<<(52-5); double s=asdouble(t); const double*Cc=exp2f_poly;
  double z=Cc[0]*r+Cc[1], r2=r*r, y=Cc[2]*r+1; y=z*r2+y; y=y*s; return (float)y; }
// powf(x, y) for finite, non-zero, non-integer y with |y*log2(x)| < 126 (y = 2.2f or 1/2.2f here)
static inline float pt_powf_host(float x,float y){
  uint32_t ix=asuint(x);
  if(ix-0x00800000u>=0x7f800000u-0x00800000u){
    if(2*ix==0||2*ix>=2u*0x7f800000u){ if(2*ix>2u*0x7f800000u) return x+y; float x2=x*x; return x2; }  // zero, inf, nan (y>0)
    if(ix&0x80000000u) return (x-x)/(x-x);  // finite x<0, non-integer y: invalid
    if(ix<0x00800000u){ ix=asuint(x*0x1p23f); ix&=0x7fffffffu; ix-=23u<<23; } }
  double logx=powf_log2_inline(ix); double ylogx=(double)y*logx; return powf_exp2_inline(ylogx,0); }
