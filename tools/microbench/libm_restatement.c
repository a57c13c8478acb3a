/* Exhaustive check (CPU, ~10 s on 8 threads) that the restatements of glibc 2.28 - 2.40 acosf / cosf / sinf used by the device
 * (gorder_amd/csrc/gm_math.h) and by the oracle MIRROR mode ARE the host libm: every float of [-1, 1] for acos, of [0, pi] for
 * cos and sin.   gcc -O2 -ffp-contract=off -o libm_restatement libm_restatement.c -lm -lpthread && ./libm_restatement
 * -> "mismatches: acosf 0 ... cosf 0 ... sinf 0" on glibc 2.35 (this image). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
static inline uint32_t asuint(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline float asfloat(uint32_t u){float f;memcpy(&f,&u,4);return f;}
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4,s1,s2,s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t abstop12(float x){return (asuint(x)>>20)&0x7ff;}
static inline float sinf_poly(double x,double x2,const sincos_t*p,int n){
  double x3,x4,x5,x6,x7,s,c,c1,c2,s1; (void)x5;
  if((n&1)==0){x3=x*x2;s1=p->s2+x2*p->s3;x7=x3*x2;s=x+x3*p->s1;return s+x7*s1;}
  else{x4=x2*x2;c2=p->c3+x2*p->c4;c1=p->c0+x2*p->c1;x6=x4*x2;c=c1+x4*p->c2;return c+x6*c2;}
}
static inline double reduce_fast(double x,const sincos_t*p,int*np){double r=x*p->hpi_inv;int n=((int32_t)r+0x800000)>>24;*np=n;return x-n*p->hpi;}
float my_cosf(float y){ double x=y,s; int n; const sincos_t*p=&T[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){double x2=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return 1.0f; return sinf_poly(x,x2,p,1);}
  else if(abstop12(y)<abstop12(120.0f)){x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n^1);}
  return cosf(y);}
float my_sinf(float y){ double x=y,s; int n; const sincos_t*p=&T[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){s=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return y; return sinf_poly(x,s,p,0);}
  else if(abstop12(y)<abstop12(120.0f)){x=reduce_fast(x,p,&n); s=p->sign[n&3]; if(n&2)p=&T[1]; return sinf_poly(x*s,x*x,p,n);}
  return sinf(y);}
static const float one=1.0f, pi=3.1415925026e+00f, pio2_hi=1.5707962513e+00f, pio2_lo=7.5497894159e-08f,
 pS0=1.6666667163e-01f,pS1=-3.2556581497e-01f,pS2=2.0121252537e-01f,pS3=-4.0055535734e-02f,pS4=7.9153501429e-04f,pS5=3.4793309169e-05f,
 qS1=-2.4033949375e+00f,qS2=2.0209457874e+00f,qS3=-6.8828397989e-01f,qS4=7.7038154006e-02f;
float my_acosf(float x){ float z,p,q,r,w,s,c,df; int32_t hx=(int32_t)asuint(x), ix=hx&0x7fffffff;
  if(ix==0x3f800000){ if(hx>0) return 0.0f; else return pi+(float)2.0*pio2_lo; }
  else if(ix>0x3f800000) return (x-x)/(x-x);
  if(ix<0x3f000000){ if(ix<=0x23000000) return pio2_hi+pio2_lo; z=x*x;
    p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; return pio2_hi-(x-(pio2_lo-x*r)); }
  else if(hx<0){ z=(one+x)*(float)0.5; p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); s=sqrtf(z); r=p/q; w=r*s-pio2_lo; return pi-(float)2.0*(s+w); }
  else { int32_t idf; z=(one-x)*(float)0.5; s=sqrtf(z); df=s; idf=(int32_t)asuint(df); df=asfloat((uint32_t)(idf&0xfffff000)); c=(z-df*df)/(s+df);
    p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; w=r*s+c; return (float)2.0*(df+w); } }
typedef struct { uint32_t lo, hi; uint64_t bad[3]; uint32_t first[3]; } job;
static void *run(void *a){ job*j=a; for(uint64_t u=j->lo; u<j->hi; u++){ float x=asfloat((uint32_t)u);
   /* acos on [-1,1]; cos and sin on [0, pi] */
   float ax=fabsf(x);
   if(ax<=1.0f){ float r0=acosf(x), r1=my_acosf(x); if(asuint(r0)!=asuint(r1)){ if(!j->bad[0]++) j->first[0]=(uint32_t)u; } }
   if(x>=0.0f && x<=3.1415927f){ float r0=cosf(x), r1=my_cosf(x); if(asuint(r0)!=asuint(r1)){ if(!j->bad[1]++) j->first[1]=(uint32_t)u; }
      r0=sinf(x); r1=my_sinf(x); if(asuint(r0)!=asuint(r1)){ if(!j->bad[2]++) j->first[2]=(uint32_t)u; } } }
  return NULL; }
int main(){ enum{N=8}; pthread_t th[N]; job jb[N]; memset(jb,0,sizeof jb);
  /* all floats of either sign up to 4.0 in magnitude cover [-1,1] and [0,pi] */
  uint64_t ranges[2][2]={{0u,0x40800000u},{0x80000000u,0xbf800001u}};
  uint64_t tot[3]={0,0,0}; uint32_t first[3]={0,0,0};
  for(int rg=0; rg<2; rg++){ uint64_t lo=ranges[rg][0], hi=ranges[rg][1], step=(hi-lo+N-1)/N;
    for(int t=0;t<N;t++){ memset(&jb[t],0,sizeof(job)); jb[t].lo=(uint32_t)(lo+t*step); uint64_t h=lo+(t+1)*step; if(h>hi)h=hi; jb[t].hi=(uint32_t)h; if(h>0xffffffffull) jb[t].hi=0xffffffffu; pthread_create(&th[t],NULL,run,&jb[t]); }
    for(int t=0;t<N;t++){ pthread_join(th[t],NULL); for(int k=0;k<3;k++){ if(jb[t].bad[k] && !tot[k]) first[k]=jb[t].first[k]; tot[k]+=jb[t].bad[k]; } } }
  printf("mismatches: acosf %llu (first %08x)  cosf %llu (first %08x)  sinf %llu (first %08x)\n",(unsigned long long)tot[0],first[0],(unsigned long long)tot[1],first[1],(unsigned long long)tot[2],first[2]);
  return 0; }
