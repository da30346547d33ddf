// oracle/stl_probe.cpp -- TEST INFRASTRUCTURE ONLY.
// Exercises the REAL libstdc++ std::unordered_set<int> of the host toolchain, the one third-party
// behaviour the reference's output depends on (call site: reference src/sampler.cpp:55,69,79).
//   stl_probe chain          -> prints the bucket-count chain of a set grown by single inserts
//   stl_probe order < seqs   -> each input line "len x0 x1 ..." ; prints the iteration order per line
#include <cstdio>
#include <cstring>
#include <unordered_set>
#include <vector>
int main(int argc, char **argv) {
    if (argc > 1 && !std::strcmp(argv[1], "chain")) {
        std::unordered_set<int> s;
        size_t last = s.bucket_count();
        for (int i = 0; i < 3000000; ++i) {
            s.insert(i);
            if (s.bucket_count() != last) { last = s.bucket_count(); std::printf("%d %zu\n", i + 1, last); }
        }
        return 0;
    }
    long len;
    while (std::scanf("%ld", &len) == 1) {
        std::unordered_set<int> s;
        for (long i = 0; i < len; ++i) { int x; if (std::scanf("%d", &x) != 1) return 1; s.insert(x); }
        std::printf("%zu", s.size());
        for (int x : s) std::printf(" %d", x);
        std::printf("\n");
    }
    return 0;
}
