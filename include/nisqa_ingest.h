/* nisqa_ingest.h -- C ABI of libnisqa_ingest.so: the host ingest either side of the hot path
 * (SURVEY.md section 8f-2).  Plain C, no HIP: it fills caller-owned (page-locked) staging memory; the H2D copy
 * and everything after it belong to libnisqa_hip.so.
 *
 * Replaces, for RIFF/WAVE and FLAC input, what the reference does per file inside
 * SpeechQualityDataset._load_spec -> get_librosa_melspec -> lb.load(path, sr=None)
 * (nisqa/NISQA_lib.py:2129-2160, 2299-2306) from the DataLoader workers of predict_mos / predict_dim
 * (NISQA_lib.py:1425-1431, 1446-1452), each decoding one file into a fresh float32 array: here a pool of native
 * threads parses the headers of a whole batch, the caller fixes the batch layout from the headers alone, and the same threads pread() every data chunk straight into its slot of the
 * staging buffer -- one host copy per sample, no interpreter lock, no per-file allocation.
 *
 * Reference-side binding (ctypes):
 *     L = ctypes.CDLL('libnisqa_ingest.so')
 *     L.nisqa_ingest_probe(paths, n, infos, threads); L.nisqa_ingest_read(paths, n, infos, dst, dst_off, threads)
 */
#ifndef NISQA_INGEST_H
#define NISQA_INGEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NISQA_INGEST_ABI_VERSION 2

/* per-file status */
#define NISQA_WAV_OK 0
#define NISQA_WAV_ERR_OPEN 1        /* cannot open / stat */
#define NISQA_WAV_ERR_FORMAT 2      /* not RIFF/WAVE, missing fmt or data chunk, bad block align, unsupported encoding */
#define NISQA_WAV_ERR_READ 3        /* short read of the data chunk */

/* WAVE format tags after resolving WAVE_FORMAT_EXTENSIBLE */
#define NISQA_WAV_TAG_PCM 1
#define NISQA_WAV_TAG_FLOAT 3
#define NISQA_WAV_TAG_ALAW 6         /* G.711 A-law, 8 bits per sample */
#define NISQA_WAV_TAG_MULAW 7        /* G.711 mu-law, 8 bits per sample */
#define NISQA_WAV_TAG_BIG_ENDIAN 0x10000   /* flag OR-ed into tag: a RIFX file -- header fields were read big-endian and the
                                              samples of the data chunk ARE big-endian (nisqa_ingest_read copies them verbatim) */

#define NISQA_WAV_TAG_FLAC 0xF1AC          /* not a WAVE file: a native FLAC stream ("fLaC", possibly behind an ID3v2 tag), 4-24 bits per
                                              sample; data_offset = offset of the "fLaC" marker, n_frames from STREAMINFO (or counted by
                                              decoding when STREAMINFO leaves it open), block_align = channels * (bits + 7) / 8 of the
                                              DECODED samples.  lb.load reads these through the same soundfile call as a WAVE file. */

typedef struct nisqa_wav_info {
    int32_t status;        /* NISQA_WAV_* */
    int32_t tag;           /* NISQA_WAV_TAG_PCM | _FLOAT | _ALAW | _MULAW, possibly | NISQA_WAV_TAG_BIG_ENDIAN */
    int32_t channels;
    int32_t bits;          /* 1..32 (PCM, in a container of (bits + 7) / 8 bytes) / 32, 64 (float) / 8 (A-law, mu-law) */
    int32_t block_align;   /* channels * bytes per sample */
    int32_t sample_rate;
    int64_t data_offset;   /* byte offset of the data chunk body in the file */
    int64_t n_frames;      /* data bytes / block_align (data size clamped to the file size; 0xFFFFFFFF = to EOF) */
} nisqa_wav_info;

int nisqa_ingest_abi_version(void);

/* Parse the RIFF headers of paths[0..n) on up to n_threads pool threads (n_threads <= 0: one per online CPU,
 * capped at 64).  info[i].status says whether file i is usable.  Returns the number of files with status != OK. */
int nisqa_ingest_probe(const char* const* paths, int32_t n, nisqa_wav_info* info, int32_t n_threads);

/* Copy the data chunk of every file i with dst_off[i] >= 0 (info[i].n_frames * info[i].block_align bytes, verbatim)
 * to (char*)dst + dst_off[i].  info must come from nisqa_ingest_probe on the same paths; failures are recorded in
 * info[i].status (NISQA_WAV_ERR_OPEN / NISQA_WAV_ERR_READ).  Returns the number of failed files.
 * A FLAC file (tag NISQA_WAV_TAG_FLAC) is DECODED into its slot as int16 -- mono 16-bit streams only (n_frames * 2 bytes, the
 * form a mono PCM16 data chunk has); any other FLAC stream fails with NISQA_WAV_ERR_FORMAT here and goes through
 * nisqa_ingest_decode_flac.  Every frame CRC, the stream length and STREAMINFO's MD5 of the samples are verified. */
int nisqa_ingest_read(const char* const* paths, int32_t n, nisqa_wav_info* info, void* dst,
                      const int64_t* dst_off, int32_t n_threads);

/* Decode one FLAC file (info from nisqa_ingest_probe, tag NISQA_WAV_TAG_FLAC) to interleaved int32 samples:
 * dst[frame * channels + channel], info->n_frames * info->channels values, each the stream's integer sample (soundfile's
 * float32 is that value / 2^(bits - 1)).  Returns NISQA_WAV_OK or an NISQA_WAV_ERR_* code. */
int nisqa_ingest_decode_flac(const char* path, const nisqa_wav_info* info, int32_t* dst);

#ifdef __cplusplus
}
#endif
#endif
