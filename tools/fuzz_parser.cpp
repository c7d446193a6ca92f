// tools/fuzz_parser.cpp -- the decoder peer's host parser (media_amd/csrc/h264_parse.h) under AddressSanitizer + UBSan:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I media_amd/csrc tools/fuzz_parser.cpp -o /tmp/fuzz_parser
//   /tmp/fuzz_parser units.bin        (units.bin: access units as <u32 length><bytes>; tests/test_dec_parser.py writes damaged ones)
// Every unit is parsed from an exact-size heap copy, alternating between the parser's two picture buffers; a read or write outside
// any array, a signed overflow or a bad shift aborts the run.  CPU only (sanitizers cannot run on the GPU pool).
#include "h264_parse.h"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    h264dec::Parser P;
    uint32_t n; size_t ok = 0, bad = 0, none = 0; int k = 0;
    std::vector<uint8_t> b;
    while (fread(&n, 4, 1, f) == 1) {
        b.resize(n);
        if (n && fread(b.data(), 1, n, f) != n) break;
        P.select(k++ & 1);
        std::vector<uint8_t> tight(b);   // exact-size heap copy: reads past the end are caught
        const int rc = P.parse_access_unit(tight.data(), tight.size());
        if (rc == 1) { ok++; volatile int s = 0; const auto& pic = P.picture(); for (auto v : pic.mbqp) s += v; for (auto v : pic.refq) s += v; } else if (rc == 0) none++; else bad++;
    }
    printf("parsed %zu refused %zu no-picture %zu\n", ok, bad, none);
}
