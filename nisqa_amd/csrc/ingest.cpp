// libnisqa_ingest.so -- native WAV ingest for the predict loop (include/nisqa_ingest.h).
//
// Replaces the per-file lb.load of the reference's DataLoader workers (nisqa/NISQA_lib.py:2299-2306) for RIFF/WAVE
// input.  A persistent pool of threads walks the RIFF chunks of a whole batch (probe) and then pread()s every data
// chunk straight into the caller's staging buffer (read): one host copy per sample -- page cache -> page-locked
// memory -- with no interpreter in the loop.  Decoding of the WAVE sample format is NOT done here: mono PCM16 goes to
// the GPU verbatim (nisqa_pcm16_to_f32 scales it there); everything else is decoded by the host mirror
// (nisqa_amd/wavio.py) with lb.load's semantics.  FLAC files (flac.hpp) ARE decoded here, by the same threads: a mono
// 16-bit stream straight into its int16 slot of the staging buffer (the same fast path as PCM16 from then on), any other
// stream to interleaved int32 for the host mirror (nisqa_ingest_decode_flac).
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <mutex>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/nisqa_ingest.h"
#include "flac.hpp"

namespace {

// ---- a small persistent pool: run(job, n) executes job(i) for i in [0, n) on up to `want` threads + the caller ----
class Pool {
public:
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    template <class F>
    void run(int n, int want, F&& job) {
        std::lock_guard<std::mutex> serial(run_m_);              // one batch at a time
        if (n <= 0) return;
        int helpers = want - 1;
        if (helpers > n - 1) helpers = n - 1;
        grow(helpers);
        std::function<void(int)> fn = job;
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            next_.store(0);
            n_ = n;
            active_ = helpers;
            allowed_ = helpers;
            ++epoch_;
        }
        if (helpers > 0) cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [&] { return active_ == 0; });
        fn_ = nullptr;
    }

private:
    void grow(int helpers) {
        while ((int)threads_.size() < helpers) {
            const int id = (int)threads_.size();
            threads_.emplace_back([this, id] { loop(id); });
        }
    }
    void work() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return quit_ || (epoch_ != seen && id < allowed_); });
                if (quit_) return;
                seen = epoch_;
            }
            work();
            {
                std::lock_guard<std::mutex> g(m_);
                if (--active_ == 0) done_cv_.notify_one();
            }
        }
    }
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, active_ = 0, allowed_ = 0;
    uint64_t epoch_ = 0;
    bool quit_ = false;
};

Pool& pool() {
    static Pool* p = new Pool();                                   // leaked on purpose: no join at process exit
    return *p;
}

int thread_count(int requested) {
    if (requested > 0) return requested > 256 ? 256 : requested;
    long c = sysconf(_SC_NPROCESSORS_ONLN);
    if (c < 1) c = 1;
    return c > 64 ? 64 : (int)c;
}

inline uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t le16(const unsigned char* p) { return p[0] | (p[1] << 8); }
inline uint32_t rd32(const unsigned char* p, bool be) { return be ? (p[3] | (p[2] << 8) | (p[1] << 16) | ((uint32_t)p[0] << 24)) : le32(p); }
inline uint32_t rd16(const unsigned char* p, bool be) { return be ? (uint32_t)(p[1] | (p[0] << 8)) : le16(p); }

bool pread_full(int fd, void* dst, size_t want, off_t at) {
    char* d = (char*)dst;
    while (want > 0) {
        const ssize_t r = pread(fd, d, want, at);
        if (r <= 0) return false;
        d += r;
        at += r;
        want -= (size_t)r;
    }
    return true;
}

// A file image in memory (FLAC frames are decoded from the whole compressed file: a tenth of the size of its samples).
bool slurp(int fd, std::vector<uint8_t>& buf) {
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0 || sb.st_size > (int64_t)1 << 31) return false;
    buf.resize((size_t)sb.st_size);
    return pread_full(fd, buf.data(), buf.size(), 0);
}

// FLAC: STREAMINFO -> the same record a WAVE header gives.  A stream whose STREAMINFO does not say how long it is (total = 0:
// a piped encoder) is decoded once here to count it.
void probe_flac(int fd, int64_t fsize, const unsigned char* head, size_t got, nisqa_wav_info* out) {
    nqflac::Stream s;
    int rc = nqflac::stream_info(head, got, s);
    unsigned char far[64];
    if (rc == 1 && (int64_t)s.marker + 42 <= fsize && s.marker > 0 && pread_full(fd, far, 42, (off_t)s.marker)) {
        const size_t at = s.marker;                        // behind a long ID3v2 tag
        rc = nqflac::stream_info(far, 42, s);
        s.marker = at;
    }
    if (rc != 0) return;
    int64_t total = s.total;
    // STREAMINFO's 36-bit total is a CLAIM until frames are decoded; staging buffers are sized from it.  A frame is at least 10 bytes
    // (header 5 + CRC-8 + one subframe byte + CRC-16 + padding) and carries at most max_block samples per channel, so a file of fsize
    // bytes cannot hold more than (fsize / 10 + 1) * max_block: a larger claim is a malformed header ("Could not load file"), not a
    // 100 GB allocation.
    if (total > (fsize / 10 + 1) * (int64_t)s.max_block) return;
    if (total == 0) {
        std::vector<uint8_t> buf;
        if (!slurp(fd, buf)) { out->status = NISQA_WAV_ERR_READ; return; }
        if (nqflac::decode(buf.data(), buf.size(), s, nullptr, &total) != 0) return;
    }
    out->tag = NISQA_WAV_TAG_FLAC;
    out->channels = s.channels;
    out->bits = s.bits;
    out->block_align = s.channels * ((s.bits + 7) / 8);
    out->sample_rate = s.sample_rate;
    out->data_offset = (int64_t)s.marker;
    out->n_frames = total;
    out->status = NISQA_WAV_OK;
}

// FLAC -> samples.  as_i16: mono 16-bit stream -> int16 at dst; otherwise interleaved int32 (channel-minor) at dst.
struct PcmSink : nqflac::Sink {
    char* dst;
    bool as_i16;
    int64_t at = 0, cap;
    PcmSink(char* d, bool i16, int64_t frames) : dst(d), as_i16(i16), cap(frames) {}
    void block(const int32_t* const* chan, int ch, int count) override {
        if (at + count > cap) count = (int)(cap - at);      // never beyond the slot the header promised
        if (as_i16) {
            int16_t* w = (int16_t*)dst + at;
            for (int i = 0; i < count; ++i) w[i] = (int16_t)chan[0][i];
        } else {
            int32_t* w = (int32_t*)dst + at * ch;
            for (int i = 0; i < count; ++i)
                for (int c = 0; c < ch; ++c) *w++ = chan[c][i];
        }
        at += count;
    }
};

int decode_flac_file(const char* path, const nisqa_wav_info* info, char* dst, bool as_i16) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return NISQA_WAV_ERR_OPEN;
    std::vector<uint8_t> buf;
    const bool ok = slurp(fd, buf);
    close(fd);
    if (!ok) return NISQA_WAV_ERR_READ;
    nqflac::Stream s;
    if ((size_t)info->data_offset + 42 > buf.size() || nqflac::stream_info(buf.data() + info->data_offset, buf.size() - (size_t)info->data_offset, s) != 0)
        return NISQA_WAV_ERR_FORMAT;
    s.marker = (size_t)info->data_offset;
    if (s.channels != info->channels || s.bits != info->bits || (s.total && s.total != info->n_frames)) return NISQA_WAV_ERR_FORMAT;
    if (as_i16 && (s.channels != 1 || s.bits != 16)) return NISQA_WAV_ERR_FORMAT;
    PcmSink sink(dst, as_i16, info->n_frames);
    int64_t done = 0;
    const int rc = nqflac::decode(buf.data(), buf.size(), s, &sink, &done);
    if (rc != 0) return rc == 3 ? NISQA_WAV_ERR_READ : NISQA_WAV_ERR_FORMAT;
    return done == info->n_frames ? NISQA_WAV_OK : NISQA_WAV_ERR_READ;
}

// Walk the RIFF (or RF64, or big-endian RIFX) chunks like soundfile/libsndfile does for the cases lb.load meets: 'fmt ' (PCM, IEEE float,
// WAVE_FORMAT_EXTENSIBLE with the sub-format in the GUID's first two bytes) then 'data'; other chunks are skipped
// (word-aligned); a data size of 0xFFFFFFFF or one that overruns the file means "to end of file".
void probe_one(const char* path, nisqa_wav_info* out) {
    std::memset(out, 0, sizeof(*out));
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { out->status = NISQA_WAV_ERR_OPEN; return; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); out->status = NISQA_WAV_ERR_OPEN; return; }
    const int64_t fsize = sb.st_size;
    unsigned char head[4096];
    const ssize_t got = pread(fd, head, sizeof(head), 0);
    out->status = NISQA_WAV_ERR_FORMAT;
    if (got >= 4 && (!std::memcmp(head, "fLaC", 4) || !std::memcmp(head, "ID3", 3))) {
        probe_flac(fd, fsize, head, (size_t)got, out);
        close(fd);
        return;
    }
    const bool be = got >= 4 && !std::memcmp(head, "RIFX", 4);       // big-endian variant: header fields AND samples
    if (got >= 12 && (!std::memcmp(head, "RIFF", 4) || !std::memcmp(head, "RF64", 4) || be) && !std::memcmp(head + 8, "WAVE", 4)) {
        int64_t pos = 12;
        bool have_fmt = false;
        while (pos + 8 <= fsize) {
            unsigned char hb[8];
            const unsigned char* h = hb;
            if (pos + 8 <= got) h = head + pos;
            else if (!pread_full(fd, hb, 8, pos)) break;
            const uint32_t size = rd32(h + 4, be);
            const int64_t body = pos + 8;
            if (!std::memcmp(h, "fmt ", 4)) {
                unsigned char fb[28] = {0};
                const size_t need = size >= 28 ? 28 : 16;
                if (size < 16) break;
                if (body + (int64_t)need <= got) std::memcpy(fb, head + body, need);
                else if (!pread_full(fd, fb, need, body)) break;
                int tag = (int)rd16(fb, be);
                out->channels = (int)rd16(fb + 2, be);
                out->sample_rate = (int32_t)rd32(fb + 4, be);
                out->block_align = (int)rd16(fb + 12, be);
                out->bits = (int)rd16(fb + 14, be);
                if (tag == 0xFFFE && size >= 28) tag = (int)(rd32(fb + 24, be) & 0xFFFFu);     // Data1 of the sub-format GUID
                out->tag = tag;
                have_fmt = true;
            } else if (!std::memcmp(h, "data", 4)) {
                if (!have_fmt) break;
                int64_t dsize = size;
                if (size == 0xFFFFFFFFu || body + dsize > fsize) dsize = fsize - body;
                const int bytes = (out->bits + 7) / 8;
                // PCM: any width up to 32 bits in its container of (bits + 7) / 8 bytes (12- and 20-bit samples are left-justified in
                // 2 / 3 bytes; libsndfile reads the container)
                const bool enc_ok = (out->tag == NISQA_WAV_TAG_PCM && out->bits >= 1 && out->bits <= 32) ||
                                    (out->tag == NISQA_WAV_TAG_FLOAT && (out->bits == 32 || out->bits == 64)) ||
                                    ((out->tag == NISQA_WAV_TAG_ALAW || out->tag == NISQA_WAV_TAG_MULAW) && out->bits == 8);
                if (out->channels < 1 || out->block_align != out->channels * bytes || !enc_ok) break;
                out->data_offset = body;
                out->n_frames = dsize / out->block_align;
                if (be) out->tag |= NISQA_WAV_TAG_BIG_ENDIAN;
                out->status = NISQA_WAV_OK;
                break;
            }
            pos = body + (int64_t)size + (size & 1);
        }
    }
    close(fd);
}

void read_one(const char* path, nisqa_wav_info* info, char* dst) {
    if (info->status != NISQA_WAV_OK) return;
    if (info->tag == NISQA_WAV_TAG_FLAC) {                   // (only a mono 16-bit stream has a verbatim int16 form)
        info->status = decode_flac_file(path, info, dst, true);
        return;
    }
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { info->status = NISQA_WAV_ERR_OPEN; return; }
    const size_t want = (size_t)info->n_frames * (size_t)info->block_align;
    if (!pread_full(fd, dst, want, (off_t)info->data_offset)) info->status = NISQA_WAV_ERR_READ;
    close(fd);
}

}  // namespace

extern "C" int nisqa_ingest_abi_version(void) { return NISQA_INGEST_ABI_VERSION; }

extern "C" int nisqa_ingest_probe(const char* const* paths, int32_t n, nisqa_wav_info* info, int32_t n_threads) {
    if (n < 0 || (n > 0 && (!paths || !info))) return -1;
    pool().run(n, thread_count(n_threads), [&](int i) { probe_one(paths[i], info + i); });
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += info[i].status != NISQA_WAV_OK;
    return bad;
}

extern "C" int nisqa_ingest_read(const char* const* paths, int32_t n, nisqa_wav_info* info, void* dst,
                                 const int64_t* dst_off, int32_t n_threads) {
    if (n < 0 || (n > 0 && (!paths || !info || !dst || !dst_off))) return -1;
    pool().run(n, thread_count(n_threads), [&](int i) {
        if (dst_off[i] >= 0) read_one(paths[i], info + i, (char*)dst + dst_off[i]);
    });
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += dst_off[i] >= 0 && info[i].status != NISQA_WAV_OK;
    return bad;
}

extern "C" int nisqa_ingest_decode_flac(const char* path, const nisqa_wav_info* info, int32_t* dst) {
    if (!path || !info || !dst || info->status != NISQA_WAV_OK || info->tag != NISQA_WAV_TAG_FLAC) return NISQA_WAV_ERR_FORMAT;
    return decode_flac_file(path, info, (char*)dst, false);
}
