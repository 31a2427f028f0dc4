// tools/bench_matgen.cpp [n_per_row] -- SdigEncoding::new host cost (matgen): C3's row length by default, 1048576 = the
// reference's matgen_bench (lcpc-brakedown-pc/src/bench.rs:22-30): clang++ -O3 -std=c++17 -pthread -Ilcpc_amd/csrc tools/bench_matgen.cpp lcpc_amd/csrc/encoding.cpp lcpc_amd/csrc/host_crypto.cpp
#include "encoding.h"
#include "host_crypto.h"
#include <chrono>
#include <stdio.h>
using namespace lcpc;
#include <stdlib.h>
int main(int argc, char** argv) {
  const uint64_t npr = argc > 1 ? strtoull(argv[1], nullptr, 10) : 166292;
  const FieldDesc& f = *field_desc(3);
  SdigSpec s; sdig_spec(3, &s);
  for (int rep = 0; rep < 3; rep++) {
    std::vector<CsrMatrix> pre, post; std::vector<LevelDims> pd, qd;
    auto t0 = std::chrono::steady_clock::now();
    sdig_generate(f, s, npr, 0, pre, post, pd, qd);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t nnz = 0; for (auto& m : pre) nnz += m.colidx.size(); for (auto& m : post) nnz += m.colidx.size();
    uint64_t chk = 0; for (auto& m : pre) for (size_t i = 0; i < m.vals.size(); i += 97) chk ^= m.vals[i] + m.colidx[i / 4 % m.colidx.size()];
    printf("sdig_generate(n_per_row=%llu): %.3f s, nnz %zu, levels %zu, chk %llx\n", (unsigned long long)npr, dt, nnz, pre.size(), (unsigned long long)chk);
    if (rep == 0) for (size_t i = 0; i < pd.size(); i++) printf("  level %zu: pre n=%llu m=%llu d=%llu | post n=%llu m=%llu d=%llu\n", i, (unsigned long long)pd[i].n, (unsigned long long)pd[i].m, (unsigned long long)pd[i].d, (unsigned long long)qd[i].n, (unsigned long long)qd[i].m, (unsigned long long)qd[i].d);
  }
}
