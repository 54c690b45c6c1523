// Mutation fuzzer for the host-side parsers that take untrusted files: the PVM/DDS decoder
// (readPVMvolume path: decodeDDS + parsePVM) and the .raw.inf sidecar parser.  Built with
// -fsanitize=address,undefined against csrc/volume_io.cpp (host only, no HIP):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -I../../volume-renderer_amd/csrc \
//       fuzz_volume_io.cpp ../../volume-renderer_amd/csrc/volume_io.cpp -o fuzz_volume_io
//   ./fuzz_volume_io <iterations> <seed file>...
// Any crash / sanitizer report is a bug; "rejected" inputs are the expected outcome.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "volume_io.h"

static std::vector<uint8_t> slurp(const char *p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s iterations seed...\n", argv[0]); return 2; }
    const long iters = std::atol(argv[1]);
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 2; i < argc; i++) seeds.push_back(slurp(argv[i]));
    std::mt19937_64 rng(12345);
    long accepted = 0, rejected = 0;
    const std::string tmp = "fuzz_volume_io.tmp";     // in the working directory
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> d = seeds[rng() % seeds.size()];
        if (d.empty()) continue;
        const int nmut = 1 + (int)(rng() % 6);
        for (int m = 0; m < nmut; m++) {
            switch (rng() % 6) {
            case 0: d[rng() % d.size()] ^= (uint8_t)(1u << (rng() % 8)); break;                       // bit flip
            case 1: d[rng() % d.size()] = (uint8_t)rng(); break;                                       // byte
            case 2: d.resize(rng() % (d.size() + 1)); if (d.empty()) d.push_back(0); break;            // truncate
            case 3: { size_t p = rng() % d.size(); d.insert(d.begin() + (long)p, (size_t)(rng() % 9), (uint8_t)rng()); } break;   // insert
            case 4: { size_t p = rng() % d.size(), n = rng() % 16; for (size_t k = 0; k < n && p + k < d.size(); k++) d[p + k] = 0xff; } break;
            default: { size_t p = rng() % d.size(); const char *num[] = {"0", "-1", "4294967295", "65536", "99999999999", "1e9"};   // header numbers
                       const char *s = num[rng() % 6]; for (size_t k = 0; s[k] && p + k < d.size(); k++) d[p + k] = (uint8_t)s[k]; } break;
            }
        }
        // in-memory entry points
        std::vector<uint8_t> out;
        std::string err;
        bool ok = vr::decodeDDS(d.data(), d.size(), out, err);
        vr::PvmVolume vol;
        if (ok) ok = vr::parsePVM(out, vol, err);
        else { std::string e2; vr::parsePVM(d, vol, e2); }                 // raw (unencoded) PVM header path
        // file entry points
        { std::ofstream f(tmp, std::ios::binary); f.write((const char *)d.data(), (std::streamsize)d.size()); }
        vr::PvmVolume v2;
        std::string e3;
        const bool ok2 = vr::readPVMvolume(tmp, v2, e3);
        vr::RawInf inf;
        std::string t, m;
        vr::parseRawInf(tmp, inf, t, m);
        (ok || ok2) ? accepted++ : rejected++;
        if (ok2 && (size_t)v2.width * v2.height * v2.depth * v2.components != v2.data.size()) {
            std::fprintf(stderr, "size mismatch: %ux%ux%ux%u vs %zu bytes\n", v2.width, v2.height, v2.depth, v2.components, v2.data.size());
            return 1;
        }
    }
    std::printf("fuzz_volume_io: %ld inputs, %ld accepted, %ld rejected, no crash\n", iters, accepted, rejected);
    return 0;
}
