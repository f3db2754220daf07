#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <cstdint>
extern "C" int kprn_host_batch_index(const int32_t*, int32_t,int32_t,int32_t,int32_t,int32_t,int32_t,int32_t,int32_t,int32_t,int32_t,int32_t*,int32_t*,int32_t*,int32_t*,int32_t*,int32_t*,int32_t*,int32_t*,int64_t*);
int main(int argc,char**argv){ int th=atoi(argv[1]); int B=16384,P=4,T=6,F=3; int64_t N=B*P, ns=N*T; std::vector<int32_t> idx(ns*F);
 srand(1); for(int64_t n=0;n<N;++n){ int pad=(rand()%100<73)?2:0; for(int t=0;t<T;++t){ int32_t*f=&idx[(n*T+t)*F]; if(t<pad){f[0]=5;f[1]=2851220;f[2]=8;} else {f[0]=1+rand()%4; f[1]=1+rand()%2851218; f[2]=1+rand()%6;} } }
 std::vector<int32_t> a(ns*F),b(N),c(N),d(N/64+1),e(24),k(ns+8),p(ns+8),u(ns+12); int64_t s[4];
 for(int r=0;r<4;++r){ auto t0=std::chrono::steady_clock::now(); kprn_host_batch_index(idx.data(),B,P,T,F,1,6,2851220,9,1,th,a.data(),b.data(),c.data(),d.data(),e.data(),k.data(),p.data(),u.data(),s); auto t1=std::chrono::steady_clock::now(); printf("threads %d: %.2f ms uniq %ld exec %ld\n",th,std::chrono::duration<double,std::milli>(t1-t0).count(),(long)s[2],(long)s[3]); } }
