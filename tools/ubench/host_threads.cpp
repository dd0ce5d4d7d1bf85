#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
int main(){ for(int nt: {1,2,4,8}){ auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> th; volatile double sink[16]; for(int t=0;t<nt;t++) th.emplace_back([t,&sink]{ double x=0; for(long i=0;i<200000000;i++) x+=i*1e-9; sink[t]=x;}); for(auto&x:th)x.join(); printf("%d threads: %.3f s\n", nt, std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count()); } }
