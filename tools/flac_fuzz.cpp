// Memory-safety fuzz of the FLAC decoder (nisqa_amd/csrc/flac.hpp) under AddressSanitizer + UBSan: bit flips, byte runs, truncations and
// scattered bytes in valid streams (written by tests/flac_enc.py); the decoder must refuse or decode, never read or write out of bounds.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -o /tmp/flac_fuzz tools/flac_fuzz.cpp
//   /tmp/flac_fuzz a.flac b.flac ...          (3 000 mutations per file; prints accepted / refused)
#include "../nisqa_amd/csrc/flac.hpp"
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

struct Sum : nqflac::Sink {
    long n = 0;
    void block(const int32_t* const* c, int ch, int count) override {
        for (int i = 0; i < count; ++i) n += c[ch - 1][i];
    }
};

int main(int argc, char** argv) {
    long ok = 0, refused = 0;
    for (int f = 1; f < argc; ++f) {
        const int fd = open(argv[f], O_RDONLY);
        struct stat sb;
        if (fd < 0 || fstat(fd, &sb) != 0) return 1;
        std::vector<uint8_t> orig((size_t)sb.st_size);
        if (read(fd, orig.data(), orig.size()) != (ssize_t)orig.size()) return 1;
        close(fd);
        srand(f);
        for (int it = 0; it < 3000; ++it) {
            std::vector<uint8_t> buf = orig;
            const size_t pos = 4 + rand() % (buf.size() - 4);
            switch (it % 4) {
                case 0: buf[pos] ^= 1 << (rand() % 8); break;
                case 1: for (int k = 0, n = 1 + rand() % 16; k < n && pos + k < buf.size(); ++k) buf[pos + k] = (uint8_t)rand(); break;
                case 2: buf.resize(pos); break;
                default: for (int k = 0; k < 8; ++k) buf[4 + rand() % (buf.size() - 4)] = (uint8_t)rand();
            }
            nqflac::Stream s;
            if (nqflac::stream_info(buf.data(), buf.size(), s) != 0) { ++refused; continue; }
            Sum sink;
            int64_t done = 0;
            if (nqflac::decode(buf.data(), buf.size(), s, &sink, &done) == 0) ++ok; else ++refused;
        }
    }
    printf("accepted %ld refused %ld\n", ok, refused);
    return 0;
}
