// Mutation fuzz of the scene-file loader (development aid): random payload bytes of a valid scene file are overwritten -- biased
// towards small integers, sign bits and huge values, the things an index or a count can be -- the header's CRC is brought up to
// date so that the damage reaches validate_flat_scene(), and whatever loads is walked by the host code that runs after a load
// (the light matching).  Built with ASan + UBSan: a validator gap shows up as an out-of-bounds read.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -static-libasan -I../../include -I../../raytracer_amd/csrc/host \
//       scene_fuzz.cpp ../../raytracer_amd/csrc/host/scene_io.cpp ../../raytracer_amd/csrc/host/flatten.cpp -lz -lpthread
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scene_io.h"

static unsigned long long s = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *f = std::fopen(argv[1], "rb"); if (!f) return 2;
    std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> orig((size_t)n); if (std::fread(orig.data(), 1, (size_t)n, f) != (size_t)n) return 2; std::fclose(f);
    const int iters = std::atoi(argv[2]); const char *tmp = argv[3];
    int loaded = 0, refused = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<unsigned char> b = orig;
        const int k = 1 + (int)(rnd() % 6);
        for (int j = 0; j < k; j++) {
            const size_t pos = 64 + (size_t)(rnd() % (unsigned)(n - 64 - 4)) / 4 * 4;          // a 4-byte word of the payload
            unsigned v;
            switch (rnd() % 6) { case 0: v = rnd() % 8; break; case 1: v = 0xffffffffu; break; case 2: v = 0x80000000u; break; case 3: v = 0x7fffffffu; break;
                                 case 4: v = rnd() % 100000; break; default: v = rnd() * 2654435761u; }
            std::memcpy(b.data() + pos, &v, 4);
        }
        const unsigned crc = (unsigned)crc32(0L, b.data() + 64, (uInt)(n - 64));                  // header: magic[8] version n_sections payload_bytes crc
        std::memcpy(b.data() + 24, &crc, 4);
        FILE *o = std::fopen(tmp, "wb"); if (!o) return 2; std::fwrite(b.data(), 1, b.size(), o); std::fclose(o);
        FlatScene fs; std::string err;
        if (load_flat_scene(tmp, fs, err)) { loaded++; mark_small_lights(fs); } else refused++;
    }
    std::printf("loaded %d, refused %d\n", loaded, refused);
    return 0;
}
