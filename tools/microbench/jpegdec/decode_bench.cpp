// decode_bench.cpp -- host Huffman decoder alone (fl_jpeghuff.cpp), best and last of 25 rounds of 8 decodes:
//   g++ -O3 -std=c++17 -Ifanlin-rs_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/microbench/jpegdec/decode_bench.cpp fanlin-rs_amd/csrc/fl_jpeghuff.cpp -o /tmp/decode_bench && /tmp/decode_bench a.jpg [b.jpg ...]
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstring>
#include "fl_jpegdec.h"
using namespace fl;
int main(int argc, char **argv)
{
    std::vector<std::vector<uint8_t>> files;
    for (int i = 1; i < argc; ++i) {
        FILE *f = fopen(argv[i], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> d(n); fread(d.data(), 1, n, f); fclose(f); files.push_back(d);
    }
    std::vector<uint8_t> blob(32 << 20);
    double best = 1e9;
    for (int rep = 0; rep < 25; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        size_t used = 0; int rc = 0; uint64_t h = 0;
        const int N = 8;
        for (int k = 0; k < N; ++k) { auto &d = files[k % files.size()]; rc |= jpeg_entropy_decode(d.data(), d.size(), blob.data(), blob.size(), &used); }
        auto t1 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < used; ++i) h = h * 1099511628211ull + blob[i];
        best = std::min(best, std::chrono::duration<double, std::milli>(t1 - t0).count() / N);
        if (rep == 24) printf("best %.3f ms; %.3f ms per file, rc %d, used %zu, hash %016llx\n", best, std::chrono::duration<double, std::milli>(t1 - t0).count() / N, rc, used, (unsigned long long)h);
    }
}
