// tools/bench_transcript.cpp -- host transcript speed: Keccak-f[1600] and the n_per_row coefficient absorbs of prove / verify
// (lcpc-2d/src/lib.rs:1045-1047).  clang++ -O3 -std=c++17 -Ilcpc_amd/csrc tools/bench_transcript.cpp lcpc_amd/csrc/host_crypto.cpp -o /tmp/bt
#include "host_crypto.h"
#include <chrono>
#include <stdio.h>
#include <vector>
#include <string.h>
using namespace lcpc;
int main() {
  uint64_t st[25]; for (int i=0;i<25;i++) st[i]=i*0x9e3779b97f4a7c15ull;
  auto t0=std::chrono::steady_clock::now();
  const int N=200000;
  for (int i=0;i<N;i++) keccak_f1600(st);
  double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  printf("keccak_f1600: %.1f ns/perm (x=%llx)\n", dt/N*1e9, (unsigned long long)st[3]);
  Transcript tr((const uint8_t*)"bench", 5);
  std::vector<uint8_t> msgs(131072*32);
  for (size_t i=0;i<msgs.size();i++) msgs[i]=(uint8_t)(i*131+7);
  t0=std::chrono::steady_clock::now();
  tr.append_messages((const uint8_t*)"$l//PR", 6, msgs.data(), 32, 131072);
  dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
  uint8_t out[32]; tr.challenge_bytes((const uint8_t*)"x",1,out,32);
  printf("append_messages 131072 x 32 B: %.2f ms  (chk %02x%02x%02x%02x)\n", dt*1e3, out[0],out[1],out[2],out[3]);
}
