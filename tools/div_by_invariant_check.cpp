// x / d through rd = RN(1 / d): q = x rd, two Markstein corrections (volpath_flat.h, div_by_invariant) against IEEE division.
//     g++ -O2 -mfma -ffp-contract=off -pthread tools/div_by_invariant_check.cpp -o /tmp/divt && /tmp/divt
// 600 random divisors per thread (plus the extremes of the significand), a full binade of dividends each and every other binade sampled: 0 differences.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <thread>
#include <vector>
#include <atomic>
#include <xmmintrin.h>
static inline uint32_t fb(float x){ uint32_t u; memcpy(&u,&x,4); return u;} static inline float bf(uint32_t u){ float x; memcpy(&x,&u,4); return x;}
static inline float divu(float x, float d, float rd) {
    float q = x * rd;
    float r = __builtin_fmaf(-q, d, x);
    q = __builtin_fmaf(r, rd, q);
    r = __builtin_fmaf(-q, d, x);
    q = __builtin_fmaf(r, rd, q);
    return bf(fb(q) | (fb(x) & 0x80000000u));
}
int main(int argc, char **argv){
  int T=8; std::atomic<long> bad{0}, tested{0}; std::vector<std::thread> th;
  for(int t=0;t<T;t++) th.emplace_back([&,t]{ _mm_setcsr(_mm_getcsr()|0x8040); uint64_t s=0x9e3779b97f4a7c15ull*(t+1); long b=0,n=0;
    for(int k=0;k<600;k++){ s^=s<<13; s^=s>>7; s^=s<<17; uint32_t dm = (uint32_t)(s>>20)&0x7fffff; int de = (int)((s>>50)%40) - 20; if (k<4) dm = (k==0?0:k==1?0x7ffffe:k==2?1:0x400000);
      float d = bf(((uint32_t)(127+de)<<23)|dm); if (dm==0x7fffff) continue; float rd = 1.0f/d;
      for(uint32_t m=0;m<0x800000;m++){ float x = bf((127u<<23)|m); // one binade of numerators
        float a=divu(x,d,rd), c=x/d; n++; if(fb(a)!=fb(c)){ if(b<3) printf("d=%a x=%a got %a want %a\n",d,x,a,c); b++; } }
      // other binades of x incl. tiny values like -log(1-u)
      for(int e=-30;e<=8;e+=1){ for(int j=0;j<2000;j++){ s^=s<<13; s^=s>>7; s^=s<<17; float x = bf(((uint32_t)(127+e)<<23)|((uint32_t)(s>>30)&0x7fffff)); float a=divu(x,d,rd), c=x/d; n++; if(fb(a)!=fb(c)){ if(b<3) printf("d=%a x=%a got %a want %a\n",d,x,a,c); b++; } } }
      float z=divu(-0.f,d,rd); if (fb(z)!=fb(-0.f/d)) b++; z=divu(0.f,d,rd); if (fb(z)!=fb(0.f/d)) b++;
    }
    bad+=b; tested+=n; });
  for(auto&t:th) t.join(); printf("tested %ld quotients, %ld differ from IEEE division\n", tested.load(), bad.load());
}
