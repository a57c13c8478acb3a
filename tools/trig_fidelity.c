#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
static long tick(float s){ return (long)round((double)s*1e6); }
int main(){
  double bias=0, absd=0, bias2=0; long n=0, mism=0, maxd=0, mism2=0;
  uint64_t st=12345;
  float nx=0.0f, ny=0.0f, nz=1.0f; float n2 = sqrtf((nx*nx+ny*ny)+nz*nz); float n2sq = (nx*nx+ny*ny)+nz*nz;
  for (int i=0;i<30000000;i++){
    float v[3]; for(int k=0;k<3;k++){ st=st*6364136223846793005ULL+1442695040888963407ULL; v[k]=((int32_t)(st>>32))*(0.25f/2147483648.0f); }
    float prod = (v[0]*nx+v[1]*ny)+v[2]*nz; float s2=(v[0]*v[0]+v[1]*v[1])+v[2]*v[2]; float n1=sqrtf(s2); if(n1==0) continue;
    float c = prod/(n1*n2); if(c<-1)c=-1; else if(c>1)c=1;
    float co = cosf(acosf(c));
    float s_ref = (1.5f*co*co)-0.5f;
    float s_dir = (1.5f*c*c)-0.5f;
    float q = (prod*prod)/(s2*n2sq); if (q>1) q=1;
    float s_sq = (1.5f*q)-0.5f;
    // exact
    double ce = (double)v[2]/sqrt((double)v[0]*v[0]+(double)v[1]*v[1]+(double)v[2]*v[2]); double se = 1.5*ce*ce-0.5;
    long d = tick(s_sq)-tick(s_ref); bias+=d; absd+=labs(d); if(d)mism++; if(labs(d)>maxd)maxd=labs(d);
    long d2 = tick(s_dir)-tick(s_ref); bias2+=d2; if(d2)mism2++;
    n++;
  }
  printf("sq-cos vs libm: mismatch %.2f%% mean bias %.4e ticks, mean|d| %.4f max|d| %ld\n",100.0*mism/n,bias/n,absd/n,maxd);
  printf("direct vs libm: mismatch %.2f%% mean bias %.4e\n",100.0*mism2/n,bias2/n);
  return 0;
}
