// libccsm_bam: native BGZF / BAM reader and modbam writer of the call_mods path (include/ccsm_bam.h).
// Host only.  Follows the SAM/BAM specification v1 (BGZF = gzip members with a 'BC' extra subfield, little-endian records);
// mirrors ccsmeth_amd/bamio.py + ccsmeth_amd/_bam2modbam.py, which the tests compare it with.
#define _FILE_OFFSET_BITS 64      // off_t / fseeko / ftello are 64-bit on every ABI (inputs pass 100 GiB)
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <new>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ccsm_bam.h"

namespace {

thread_local std::string g_err;
int fail(const std::string& msg) {
    g_err = msg;
    return 1;
}

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void wr16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
inline void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

constexpr size_t kBlockPayload = 0xff00;      // uncompressed bytes per BGZF block written
constexpr int kBlocksPerRound = 256;          // blocks inflated / deflated per parallel round
const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

template <typename F>
void parallel_for(int n, int threads, F f) {   // f(i) for i in [0, n), static interleaved partition
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) {
        for (int i = 0; i < n; ++i) f(i);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([=]() { for (int i = t; i < n; i += threads) f(i); });
    for (auto& th : pool) th.join();
}
template <typename F>
void parallel_slots(int n, int threads, F f) {   // f(i, slot) for i in [0, n): items are claimed dynamically, slot = the worker's number
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) {
        for (int i = 0; i < n; ++i) f(i, 0);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&next, n, t, &f]() { for (int i; (i = next.fetch_add(1)) < n;) f(i, t); });
    for (auto& th : pool) th.join();
}

// ---- DEFLATE codec: libdeflate when the runtime library is present (what htslib itself prefers: ~2-3x zlib's inflate rate and
// ~2x its level-6 deflate rate on BAM records), else zlib.  The image carries libdeflate.so.0 without its header, so the six entry
// points are resolved by name; CCSM_BAM_ZLIB=1 forces zlib.  Both produce / accept standard raw DEFLATE streams: files written
// with one are read by the other (and by htslib).
struct LibDeflate {
    void* (*alloc_dec)() = nullptr;
    int (*dec)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_dec)(void*) = nullptr;
    void* (*alloc_com)(int) = nullptr;
    size_t (*com)(void*, const void*, size_t, void*, size_t) = nullptr;
    void (*free_com)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
};
const LibDeflate& libdeflate() {
    static const LibDeflate ld = []() {
        LibDeflate d;
        const char* z = std::getenv("CCSM_BAM_ZLIB");
        if (z && z[0] == '1') return d;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return d;
        d.alloc_dec = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        d.dec = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_deflate_decompress"));
        d.free_dec = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        d.alloc_com = reinterpret_cast<void* (*)(int)>(dlsym(h, "libdeflate_alloc_compressor"));
        d.com = reinterpret_cast<size_t (*)(void*, const void*, size_t, void*, size_t)>(dlsym(h, "libdeflate_deflate_compress"));
        d.free_com = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_compressor"));
        d.crc = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        d.ok = d.alloc_dec && d.dec && d.free_dec && d.alloc_com && d.com && d.free_com && d.crc;
        return d;
    }();
    return ld;
}
inline uint32_t crc_of(const uint8_t* p, size_t n) {
    const LibDeflate& ld = libdeflate();
    return ld.ok ? ld.crc(0, p, n) : (uint32_t)crc32(0L, p, (uInt)n);
}
// One worker's inflate / deflate state (allocated lazily, reused over the blocks that worker handles)
struct Codec {
    void* dec = nullptr;
    void* com = nullptr;
    int com_level = 0;
    ~Codec() {
        const LibDeflate& ld = libdeflate();
        if (dec) ld.free_dec(dec);
        if (com) ld.free_com(com);
    }
    // raw DEFLATE `in` -> exactly `isize` bytes at out, CRC checked
    bool inflate_block(const uint8_t* in, size_t in_len, uint8_t* out, size_t isize, uint32_t crc) {
        if (isize == 0) return true;
        const LibDeflate& ld = libdeflate();
        if (ld.ok) {
            if (!dec && !(dec = ld.alloc_dec())) return false;
            size_t got = 0;
            if (ld.dec(dec, in, in_len, out, isize, &got) != 0 || got != isize) return false;
        } else {
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            zs.next_in = const_cast<uint8_t*>(in);
            zs.avail_in = (uInt)in_len;
            zs.next_out = out;
            zs.avail_out = (uInt)isize;
            const int rc = inflate(&zs, Z_FINISH);
            const size_t tot = zs.total_out;
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || tot != isize) return false;
        }
        return crc_of(out, isize) == crc;
    }
    // `len` bytes -> raw DEFLATE at out (capacity cap); 0 = failed / did not fit
    size_t deflate_block(const uint8_t* in, size_t len, uint8_t* out, size_t cap, int level) {
        const LibDeflate& ld = libdeflate();
        if (ld.ok) {
            if (com && com_level != level) { ld.free_com(com); com = nullptr; }
            if (!com && !(com = ld.alloc_com(level))) return 0;
            com_level = level;
            return ld.com(com, in, len, out, cap);
        }
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
        zs.next_in = const_cast<uint8_t*>(in);
        zs.avail_in = (uInt)len;
        zs.next_out = out;
        zs.avail_out = (uInt)cap;
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        return rc == Z_STREAM_END ? clen : 0;
    }
};

struct RawBlock {
    std::vector<uint8_t> cdata;
    uint64_t coffset = 0;          // file offset of the block's first byte
    uint32_t crc = 0, isize = 0;
    bool ok = true;
};

const char kSeqDecode[] = "=ACMGRSVTWYHKDBN";

inline uint8_t comp_fwd(uint8_t b) {   // bamio.py _COMP: A<->T, C<->G, N, everything else unchanged
    switch (b) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return b;
    }
}

// tag walker: calls f(tag0, tag1, type, sub, count, value_ptr, whole_ptr, whole_len) for every tag; false on malformed data
template <typename F>
bool walk_tags(const uint8_t* p, const uint8_t* end, F f) {
    while (p < end) {
        if (end - p < 3) return false;
        const uint8_t* start = p;
        const uint8_t t0 = p[0], t1 = p[1], typ = p[2];
        p += 3;
        uint8_t sub = 0;
        int64_t count = 1;
        const uint8_t* val = p;
        size_t sz;
        switch (typ) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': {
                const void* z = std::memchr(p, 0, (size_t)(end - p));
                if (!z) return false;
                sz = (size_t)((const uint8_t*)z - p) + 1;
                break;
            }
            case 'B': {
                if (end - p < 5) return false;
                sub = p[0];
                count = (int32_t)rd32(p + 1);
                size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
                if (es == 0 || count < 0) return false;
                val = p + 5;
                sz = 5 + es * (size_t)count;
                break;
            }
            default: return false;
        }
        if ((size_t)(end - p) < sz) return false;
        p += sz;
        f(t0, t1, typ, sub, count, val, start, (size_t)(p - start));
    }
    return true;
}

struct OwnedBatch {
    ccsm_bam_batch view;          // first member: the public pointer is the object's address
    std::vector<uint8_t> records;
    std::vector<int64_t> rec_offset, offset;
    std::vector<int32_t> flag, length, n_sites;
    std::vector<uint8_t> seq, fi, ri, fp, rp;
    std::vector<float> fn, rn;
    std::vector<uint64_t> name_hash;
};

// 64-bit FNV-1a of a read name (without its NUL): the key of the read's device-drawn initial states (ccsm_reads.h0_key), so that a
// site's probability depends on the read it sits in and its position there, not on where the read stands in the file or on which
// GPU it is processed
inline uint64_t fnv1a64(const uint8_t* p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}


struct OwnedCalls {
    ccsm_bam_modcalls view;       // first member
    std::vector<int32_t> tid, pos;
    std::vector<uint8_t> strand, ml, hap;
};

struct CallRows {
    std::vector<int32_t> tid, pos;
    std::vector<uint8_t> strand, ml, hap;
    int64_t used = 0;
};

// int(<tag value>) of the haplotype tag as call_mods_freq_bam.py:502-508 reads it: 1 / 2 kept, everything else 0
inline uint8_t hap_of(uint8_t typ, const uint8_t* v) {
    long long x = 0;
    switch (typ) {
        case 'c': x = (int8_t)v[0]; break;
        case 'C': x = v[0]; break;
        case 's': x = (int16_t)rd16(v); break;
        case 'S': x = rd16(v); break;
        case 'i': x = (int32_t)rd32(v); break;
        case 'I': x = rd32(v); break;
        case 'f': { uint32_t u = rd32(v); float f; std::memcpy(&f, &u, 4); if (!(f > -1e18f && f < 1e18f)) return 0; x = (long long)f; break; }
        case 'A': x = (v[0] >= '0' && v[0] <= '9') ? v[0] - '0' : 0; break;
        case 'Z': {
            char* e = nullptr;
            const char* s = (const char*)v;
            x = std::strtoll(s, &e, 10);
            if (e == s || *e != 0) return 0;
            break;
        }
        default: return 0;
    }
    return (x == 1 || x == 2) ? (uint8_t)x : 0;
}

// MM:Z / ML:B:C of one record -> calls (query position on SEQ as stored, ML), ascending in query position.
// false = the record yields no calls (no tags, no C+m group, or tags that do not fit the read).
bool parse_mm_ml(const char* mm, const uint8_t* ml, int64_t ml_len, const uint8_t* packed, uint32_t l_seq, bool reverse, char modbase,
                 char modification, std::vector<std::pair<int32_t, uint8_t>>& calls) {
    calls.clear();
    if (!mm || !ml) return false;
    const char* g = mm;
    int64_t ml_off = 0, my_off = -1;
    const char* mine = nullptr;       // the deltas of the C+m group (after the comma), or "" when it has none
    const char* mine_end = nullptr;
    while (*g) {
        const char* ge = std::strchr(g, ';');
        if (!ge) ge = g + std::strlen(g);
        if (ge > g) {
            // <base><strand><codes>[?.][,d,d,...]
            const char* p = g + 1;
            if (p < ge && (*p == '+' || *p == '-')) ++p; else return false;
            const char* codes = p;
            int ncodes = 0;
            if (p < ge && *p >= '0' && *p <= '9') { while (p < ge && *p >= '0' && *p <= '9') ++p; ncodes = 1; }
            else { while (p < ge && ((*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z'))) { ++p; ++ncodes; } }
            const char* codes_end = p;
            if (p < ge && (*p == '?' || *p == '.')) ++p;
            int64_t nd = 0;
            const char* deltas = p;
            if (p < ge) {
                if (*p != ',') return false;
                nd = 1;
                for (const char* q = p + 1; q < ge; ++q) nd += (*q == ',') ? 1 : 0;
                deltas = p + 1;
            }
            if (!mine && g[0] == modbase && g[1] == '+' && codes_end > codes && codes[0] == modification) {
                if (ncodes != 1) return false;          // combined codes (C+mh): outside what ccsmeth writes
                mine = deltas;
                mine_end = ge;
                my_off = ml_off;
                if (nd == 0) return false;              // _get_moddict_in_tags: no deltas -> {}
            }
            ml_off += nd * (ncodes > 0 ? ncodes : 1);
        }
        g = (*ge) ? ge + 1 : ge;
    }
    if (!mine || ml_off != ml_len) return false;
    // positions of modbase in the forward sequence; a reverse record's forward sequence is the reverse complement of SEQ
    const uint8_t want = reverse ? comp_fwd((uint8_t)modbase) : (uint8_t)modbase;
    auto base_at = [&](uint32_t i) -> uint8_t { return (uint8_t)kSeqDecode[(packed[i >> 1] >> ((i & 1) ? 0 : 4)) & 15]; };
    int64_t k = my_off;
    int64_t fpos = -1;                 // index in the forward sequence of the last matched base
    const char* p = mine;
    while (p < mine_end) {
        char* e = nullptr;
        const long long d = std::strtoll(p, &e, 10);
        if (e == p || d < 0 || (e < mine_end && *e != ',')) return false;
        p = (e < mine_end) ? e + 1 : e;
        long long skip = d;
        int64_t i = fpos + 1;
        for (; i < (int64_t)l_seq; ++i) {
            const uint8_t b = base_at(reverse ? (uint32_t)(l_seq - 1 - i) : (uint32_t)i);
            if (b == want) { if (skip == 0) break; --skip; }
        }
        if (i >= (int64_t)l_seq) return false;          // IndexError in the reference -> {}
        fpos = i;
        calls.emplace_back((int32_t)(reverse ? (int64_t)l_seq - 1 - i : i), ml[k++]);
    }
    if (reverse) std::reverse(calls.begin(), calls.end());
    return true;
}

void modcalls_of_record(const uint8_t* src, const ccsm_bam_modcall_opts& o, const uint8_t* const* site_mask, const int64_t* mask_len,
                        int32_t n_ref, std::vector<std::pair<int32_t, uint8_t>>& calls, CallRows& out) {
    const uint32_t bs = rd32(src);
    const uint8_t* body = src + 4;
    const int32_t tid = (int32_t)rd32(body), pos0 = (int32_t)rd32(body + 4);
    const uint32_t l_name = body[8], mapq = body[9], n_cig = rd16(body + 12), flag = rd16(body + 14), l_seq = rd32(body + 16);
    if (flag & (0x4 | 0x100 | 0x400)) return;                        // unmapped / secondary / duplicate (:489-490)
    if (o.no_supplementary && (flag & 0x800)) return;
    if ((int32_t)mapq < o.mapq) return;
    const uint8_t* cig = body + 32 + l_name;
    const uint8_t* packed = cig + 4 * (size_t)n_cig;
    const size_t fixed = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    int64_t cnt[16] = {0};
    for (uint32_t c = 0; c < n_cig; ++c) { const uint32_t v = rd32(cig + 4 * c); cnt[v & 15] += v >> 4; }
    {   // compute_pct_identity over "MIDNSHP=XB": everything but S and H aligns, M and = match
        const int64_t nalign = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[6] + cnt[7] + cnt[8] + cnt[9];
        const int64_t nmatch = cnt[0] + cnt[7];
        const double identity = nalign > 0 ? (double)nmatch / (double)nalign : 0.0;
        if (identity < o.identity) return;
    }
    const bool reverse = (flag & 16) != 0;
    const char* mm = nullptr;
    const uint8_t* ml = nullptr;
    int64_t ml_len = 0;
    uint8_t hap = 0;
    bool have_hap = false;
    walk_tags(body + fixed, body + bs, [&](uint8_t t0, uint8_t t1, uint8_t typ, uint8_t sub, int64_t count, const uint8_t* val, const uint8_t*, size_t) {
        if (t0 == 'M' && t1 == 'M' && typ == 'Z') { if (!mm) mm = (const char*)val; }
        else if (t0 == 'M' && t1 == 'L' && typ == 'B' && sub == 'C') { if (!ml) { ml = val; ml_len = count; } }
        else if (t0 == (uint8_t)o.hap_tag[0] && t1 == (uint8_t)o.hap_tag[1] && !have_hap) { hap = hap_of(typ, val); have_hap = true; }
    });
    out.used += 1;
    const bool have_calls = parse_mm_ml(mm, ml, ml_len, packed, l_seq, reverse, o.modbase, o.modification, calls);
    if (!have_calls) calls.clear();
    const uint8_t* mask = (o.refsites_all && site_mask && tid >= 0 && tid < n_ref) ? site_mask[tid] : nullptr;
    const int64_t mlen = mask ? mask_len[tid] : 0;
    const uint8_t mbit = reverse ? 2 : 1;
    if (calls.empty() && !mask) return;
    // aligned pairs in CIGAR order: M/=/X (q, r); with refsites_all also I/S (q, None) and D/N (None, r)
    int64_t n_pairs = cnt[0] + cnt[7] + cnt[8];
    if (o.refsites_all) n_pairs += cnt[1] + cnt[4] + cnt[2] + cnt[3];
    const int64_t lo = o.base_clip > 0 ? o.base_clip : 0;
    const int64_t hi = o.base_clip > 0 ? n_pairs - o.base_clip : n_pairs;
    int64_t idx = 0;
    int64_t q = 0, r = pos0;
    size_t j = 0;
    auto emit = [&](int64_t rp, uint8_t v) {
        out.tid.push_back(tid); out.pos.push_back((int32_t)rp); out.strand.push_back(reverse ? 1 : 0); out.ml.push_back(v); out.hap.push_back(hap);
    };
    for (uint32_t c = 0; c < n_cig && idx < hi; ++c) {
        const uint32_t v = rd32(cig + 4 * c);
        const uint32_t op = v & 15;
        const int64_t len = v >> 4;
        switch (op) {
            case 0: case 7: case 8:
                for (int64_t t = 0; t < len && idx < hi; ++t, ++idx, ++q, ++r) {
                    if (idx < lo) continue;
                    while (j < calls.size() && calls[j].first < q) ++j;
                    if (j < calls.size() && calls[j].first == q) emit(r, calls[j].second);
                    else if (mask && r >= 0 && r < mlen && (mask[r] & mbit)) emit(r, 0);
                }
                break;
            case 1: case 4:
                q += len;
                if (o.refsites_all) idx += len;
                break;
            case 2: case 3:
                if (o.refsites_all) {
                    for (int64_t t = 0; t < len && idx < hi; ++t, ++idx, ++r)
                        if (idx >= lo && mask && r >= 0 && r < mlen && (mask[r] & mbit)) emit(r, 0);
                    // idx may have stopped early; r is not used after the loop ends in that case
                } else {
                    r += len;
                }
                break;
            default: break;   // H, P, B: no pairs
        }
    }
}

}  // namespace

struct ccsm_bam_reader {
    FILE* fh = nullptr;
    std::string path;              // (ccsm_bam_eof_voffset walks the block headers through a handle of its own)
    int threads = 1;
    std::vector<uint8_t> stream;   // inflated bytes not yet consumed: [pos, size)
    size_t pos = 0;
    bool file_eof = false;
    std::string text;
    std::vector<uint8_t> refs;
    int32_t n_ref = 0;
    // virtual file offsets (BGZF: block file offset << 16 | offset in the block) of positions in `stream`
    struct BlockPos { uint64_t abs_start, coffset; uint32_t isize; };
    std::deque<BlockPos> bpos;
    uint64_t erased = 0;           // bytes dropped from the front of `stream` so far
    uint64_t next_coffset = 0;
    uint64_t limit_coffset = ~0ull; // blocks starting beyond this file offset are not read (ccsm_bam_seek with an end offset)
    uint64_t first_voffset = 0;     // virtual offset of the first record (behind the header)
    uint64_t stop_coffset = ~0ull;  // chunk mode (ccsm_bam_seek_chunk): records starting in a block at or beyond it are not returned
    std::vector<Codec> codecs;      // one per inflate worker
    uint64_t inflated = 0;         // bytes inflated so far (ccsm_bam_inflated_bytes)
    uint64_t abs_pos() const { return erased + pos; }
    // voffset of absolute inflated position p; at a block boundary: the start of the following block when there is one
    uint64_t voffset(uint64_t p) {
        while (bpos.size() > 1 && bpos.front().abs_start + bpos.front().isize <= p && bpos[1].abs_start <= p) bpos.pop_front();
        for (size_t i = 0; i < bpos.size(); ++i) {
            const BlockPos& b = bpos[i];
            if (p >= b.abs_start && p < b.abs_start + b.isize) return (b.coffset << 16) | (p - b.abs_start);
            if (p == b.abs_start + b.isize && (i + 1 == bpos.size() || bpos[i + 1].abs_start > p)) return (b.coffset << 16) | b.isize;
        }
        return (next_coffset << 16);
    }

    // read one BGZF block's compressed payload; false at clean EOF; throws through `err`
    bool read_block(RawBlock& b, std::string& err) {
        uint8_t head[12];
        b.coffset = next_coffset;
        if (next_coffset > limit_coffset) return false;               // a seek's range ends here: treated as end of file
        const size_t got = std::fread(head, 1, 12, fh);
        if (got == 0) return false;
        if (got < 12 || head[0] != 0x1f || head[1] != 0x8b || head[2] != 8 || head[3] != 4) {
            err = "not a BGZF stream (bad gzip member header)";
            return false;
        }
        const uint32_t xlen = rd16(head + 10);
        std::vector<uint8_t> extra(xlen);
        if (std::fread(extra.data(), 1, xlen, fh) != xlen) { err = "truncated BGZF block"; return false; }
        int bsize = -1;
        for (size_t off = 0; off + 4 <= xlen;) {
            const uint32_t slen = rd16(extra.data() + off + 2);
            if (extra[off] == 66 && extra[off + 1] == 67 && slen == 2) bsize = rd16(extra.data() + off + 4);
            off += 4 + slen;
        }
        if (bsize < 0) { err = "gzip member without the BGZF 'BC' field"; return false; }
        const long clen = (long)bsize - (long)xlen - 19;
        if (clen < 0) { err = "corrupt BGZF block size"; return false; }
        b.cdata.resize((size_t)clen);
        uint8_t tail[8];
        if (std::fread(b.cdata.data(), 1, (size_t)clen, fh) != (size_t)clen || std::fread(tail, 1, 8, fh) != 8) {
            err = "truncated BGZF block";
            return false;
        }
        b.crc = rd32(tail);
        b.isize = rd32(tail + 4);
        if (b.isize > 65536) { err = "corrupt BGZF block (ISIZE > 64 KiB)"; return false; }
        next_coffset += (uint64_t)bsize + 1;
        return true;
    }

    // make at least `need` unconsumed bytes available (fewer only at end of file / of the seek range).  Blocks are read and inflated
    // in rounds of up to kBlocksPerRound, one round's blocks in parallel, straight into `stream`.  Inside a seek range
    // (limit_coffset) a round takes every block that is left of the range; in chunk mode (stop_coffset) reading goes on behind the
    // chunk only as far as the record in hand needs.
    int fill(size_t need) {
        while (stream.size() - pos < need && !file_eof) {
            if (pos > 0 && pos >= stream.size() / 2) {
                stream.erase(stream.begin(), stream.begin() + (long)pos);
                erased += pos;
                pos = 0;
            }
            std::vector<RawBlock> blocks;
            blocks.reserve(kBlocksPerRound);
            std::string err;
            size_t have = stream.size() - pos;
            for (int i = 0; i < kBlocksPerRound; ++i) {
                if (next_coffset >= stop_coffset && have >= need) break;
                RawBlock b;
                if (!read_block(b, err)) {
                    if (!err.empty()) return fail(err);
                    file_eof = true;
                    break;
                }
                have += b.isize;
                blocks.push_back(std::move(b));
            }
            std::vector<size_t> at(blocks.size());
            size_t total = stream.size();
            for (size_t i = 0; i < blocks.size(); ++i) { at[i] = total; total += blocks[i].isize; }
            stream.resize(total);
            if (codecs.size() < (size_t)threads) codecs.resize((size_t)threads);
            uint8_t* base = stream.data();
            parallel_slots((int)blocks.size(), threads, [&](int i, int slot) {
                RawBlock& b = blocks[(size_t)i];
                b.ok = codecs[(size_t)slot].inflate_block(b.cdata.data(), b.cdata.size(), base + at[(size_t)i], b.isize, b.crc);
            });
            for (size_t i = 0; i < blocks.size(); ++i) {
                const RawBlock& b = blocks[i];
                if (!b.ok) return fail("BGZF block failed its CRC / size check");
                if (b.isize) bpos.push_back({erased + at[i], b.coffset, b.isize});
                inflated += b.isize;
            }
        }
        return 0;
    }
    size_t avail() const { return stream.size() - pos; }
};

struct ccsm_bam_writer {
    FILE* fh = nullptr;
    int threads = 1, level = 6;
    std::vector<uint8_t> buf;      // uncompressed bytes not yet written
    std::vector<Codec> codecs;     // one per deflate worker
    uint64_t file_off = 0;         // bytes written to the file so far
    // A "run" = the blocks between two ccsm_bam_writer_flush calls: it starts on a fresh BGZF block, and its blocks are cut every
    // kBlockPayload uncompressed bytes, so a record's virtual offsets follow from its byte offsets in the run once the compressed
    // sizes of the run's blocks are known (index tracking: ccsm_bam_writer_track_index).
    bool track = false;
    uint64_t run_upos = 0;         // uncompressed bytes put since the run began
    uint64_t run_file_start = 0;
    std::vector<uint64_t> blk_off; // file offset of every block of the run written so far
    struct Pending { int32_t tid, pos, end; uint32_t flag; uint64_t ubeg, uend; };
    std::vector<Pending> pend;     // placed records of the current run
    // finished runs since the last ccsm_bam_writer_take_index
    std::vector<ccsm_bam_index_entry> done;
    int64_t acc_records = 0, acc_unplaced = 0;
    bool acc_sorted = true, acc_any = false;
    uint64_t acc_first_k1 = 0, acc_last_k1 = 0;
    uint32_t acc_first_k2 = 0, acc_last_k2 = 0;
    int64_t acc_file_start = -1;
    std::vector<ccsm_bam_index_entry> handed;   // what the last take returned (owned here)

    int flush_blocks(bool all) {
        const size_t nfull = buf.size() / kBlockPayload;
        const size_t nblk = all ? (buf.size() + kBlockPayload - 1) / kBlockPayload : nfull;
        if (nblk == 0) return 0;
        std::vector<std::vector<uint8_t>> out(nblk);
        std::vector<char> ok(nblk, 1);
        if (codecs.size() < (size_t)threads) codecs.resize((size_t)threads);
        parallel_slots((int)nblk, threads, [&](int i, int slot) {
            const size_t beg = (size_t)i * kBlockPayload, len = std::min(kBlockPayload, buf.size() - beg);
            std::vector<uint8_t>& o = out[(size_t)i];
            o.resize(18 + 65536 + 8);
            // a block is at most 64 KiB on disk: 18 header + clen + 8 trailer <= 65536
            size_t clen = codecs[(size_t)slot].deflate_block(buf.data() + beg, len, o.data() + 18, 65536 - 26, level);
            if (clen == 0) {       // did not fit: stored blocks always do (0xff00 + 5 bytes per 65535)
                z_stream zs;
                std::memset(&zs, 0, sizeof(zs));
                if (deflateInit2(&zs, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK) {
                    zs.next_in = buf.data() + beg; zs.avail_in = (uInt)len; zs.next_out = o.data() + 18; zs.avail_out = (uInt)(65536 - 26);
                    if (deflate(&zs, Z_FINISH) == Z_STREAM_END) clen = zs.total_out;
                    deflateEnd(&zs);
                }
            }
            if (clen == 0 || clen + 26 > 65536) { ok[(size_t)i] = 0; return; }
            static const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
            std::memcpy(o.data(), head, 16);
            wr16(o.data() + 16, (uint32_t)(clen + 25));
            wr32(o.data() + 18 + clen, crc_of(buf.data() + beg, len));
            wr32(o.data() + 18 + clen + 4, (uint32_t)len);
            o.resize(18 + clen + 8);
        });
        for (size_t i = 0; i < nblk; ++i) {
            if (!ok[i]) return fail("BGZF deflate failed");
            if (std::fwrite(out[i].data(), 1, out[i].size(), fh) != out[i].size()) return fail("write failed");
            if (track) blk_off.push_back(file_off);
            file_off += out[i].size();
        }
        const size_t used = std::min(buf.size(), nblk * kBlockPayload);
        buf.erase(buf.begin(), buf.begin() + (long)used);
        return 0;
    }
    int put(const uint8_t* p, size_t n) {
        buf.insert(buf.end(), p, p + n);
        run_upos += n;
        if (buf.size() >= (size_t)kBlocksPerRound * kBlockPayload) return flush_blocks(false);
        return 0;
    }
    // the run ends here (everything is on disk): resolve its records' virtual offsets and start the next run
    void end_run() {
        if (track) {
            const uint64_t run_total = run_upos;
            auto voff = [&](uint64_t u) -> uint64_t {
                const uint64_t b = u / kBlockPayload, in = u % kBlockPayload;
                if (b < blk_off.size()) return (blk_off[(size_t)b] << 16) | in;
                // u == run_total on a block boundary: the first block of whatever follows the run
                (void)run_total;
                return file_off << 16;
            };
            for (const Pending& q : pend) done.push_back({q.tid, q.pos, q.end, q.flag, voff(q.ubeg), voff(q.uend)});
            if (acc_file_start < 0) acc_file_start = (int64_t)run_file_start;
        }
        pend.clear();
        blk_off.clear();
        run_upos = 0;
        run_file_start = file_off;
    }
    void note_record(const uint8_t* body, uint64_t ubeg, uint64_t uend);
};

// No exception may cross the C boundary: allocation failures and the like become an error code + ccsm_bam_last_error().
#define CCSM_BAM_CATCH                                                                        \
    catch (const std::bad_alloc&) { return fail("out of host memory"); }                      \
    catch (const std::exception& e) { return fail(std::string("unexpected: ") + e.what()); }

extern "C" {

const char* ccsm_bam_last_error(void) { return g_err.c_str(); }

int ccsm_bam_open(const char* path, int threads, ccsm_bam_reader** out) try {
    if (!path || !out) return fail("path and out must be non-NULL");
    *out = nullptr;
    ccsm_bam_reader* r = new (std::nothrow) ccsm_bam_reader();
    if (!r) return fail("out of memory");
    r->threads = std::max(1, threads);
    r->path = path;
    r->fh = std::fopen(path, "rb");
    if (!r->fh) { delete r; return fail(std::string("cannot open ") + path); }
    auto bail = [&](const std::string& m) { std::fclose(r->fh); delete r; return fail(m); };
    if (r->fill(12)) return bail(g_err);
    if (r->avail() < 12 || std::memcmp(r->stream.data() + r->pos, "BAM\1", 4) != 0) return bail(std::string(path) + " is not a BAM file");
    const uint32_t l_text = rd32(r->stream.data() + r->pos + 4);
    if (r->fill(12 + (size_t)l_text)) return bail(g_err);
    if (r->avail() < 12 + (size_t)l_text) return bail("truncated BAM header");
    const char* tp = reinterpret_cast<const char*>(r->stream.data() + r->pos + 8);
    r->text.assign(tp, strnlen(tp, l_text));
    r->n_ref = (int32_t)rd32(r->stream.data() + r->pos + 8 + l_text);
    r->pos += 12 + l_text;
    for (int i = 0; i < r->n_ref; ++i) {
        if (r->fill(4)) return bail(g_err);
        if (r->avail() < 4) return bail("truncated BAM reference list");
        const uint32_t l_name = rd32(r->stream.data() + r->pos);
        if (r->fill(8 + (size_t)l_name)) return bail(g_err);
        if (r->avail() < 8 + (size_t)l_name) return bail("truncated BAM reference list");
        r->refs.insert(r->refs.end(), r->stream.begin() + (long)r->pos, r->stream.begin() + (long)(r->pos + 8 + l_name));
        r->pos += 8 + l_name;
    }
    if (r->fill(1)) return bail(g_err);
    r->first_voffset = r->voffset(r->abs_pos());
    *out = r;
    return 0;
} CCSM_BAM_CATCH

int ccsm_bam_header(const ccsm_bam_reader* r, const char** text, int64_t* text_len, const uint8_t** refs, int64_t* refs_len,
                    int32_t* n_ref) {
    if (!r || !text || !text_len || !refs || !refs_len || !n_ref) return fail("arguments must be non-NULL");
    *text = r->text.data();
    *text_len = (int64_t)r->text.size();
    *refs = r->refs.data();
    *refs_len = (int64_t)r->refs.size();
    *n_ref = r->n_ref;
    return 0;
}

void ccsm_bam_close(ccsm_bam_reader* r) {
    if (!r) return;
    if (r->fh) std::fclose(r->fh);
    delete r;
}

void ccsm_bam_batch_free(ccsm_bam_batch* b) { delete reinterpret_cast<OwnedBatch*>(b); }

int ccsm_bam_next(ccsm_bam_reader* r, int32_t max_reads, ccsm_bam_batch** out) try {
    if (!r || !out) return fail("reader and out must be non-NULL");
    *out = nullptr;
    if (max_reads <= 0) return fail("max_reads must be > 0");
    OwnedBatch* ob = new (std::nothrow) OwnedBatch();
    if (!ob) return fail("out of memory");
    ob->rec_offset.push_back(0);
    if (r->fill(4)) { delete ob; return 1; }
    ob->view.voffset_start = r->voffset(r->abs_pos());
    for (int n = 0; n < max_reads; ++n) {
        if (r->fill(4)) { delete ob; return 1; }
        if (r->avail() == 0) break;
        if (r->stop_coffset != ~0ull && (r->voffset(r->abs_pos()) >> 16) >= r->stop_coffset) break;   // the chunk's last record is behind us
        if (r->avail() < 4) { delete ob; return fail("truncated BAM record"); }
        const uint32_t bs = rd32(r->stream.data() + r->pos);
        if (bs < 32) { delete ob; return fail("corrupt BAM record (block_size < 32)"); }
        if (r->fill(4 + (size_t)bs)) { delete ob; return 1; }
        if (r->avail() < 4 + (size_t)bs) { delete ob; return fail("truncated BAM record"); }
        const uint8_t* rec = r->stream.data() + r->pos;
        const uint8_t* body = rec + 4;
        const uint32_t l_name = body[8], n_cig = rd16(body + 12), flag = rd16(body + 14);
        const uint32_t l_seq = rd32(body + 16);
        const size_t fixed = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
        if (fixed > bs) { delete ob; return fail("corrupt BAM record (field lengths exceed block_size)"); }
        const uint8_t* packed = body + 32 + l_name + 4 * n_cig;
        const uint8_t* tags = body + fixed;
        const uint8_t* end = body + bs;
        const uint8_t *tfi = nullptr, *tri = nullptr, *tfp = nullptr, *trp = nullptr;
        bool has_fn = false, has_rn = false;
        float fn = 0.f, rn = 0.f;
        auto as_num = [](uint8_t typ, const uint8_t* v, float& o) -> bool {
            switch (typ) {
                case 'c': o = (float)(int8_t)v[0]; return true;
                case 'C': o = (float)v[0]; return true;
                case 's': o = (float)(int16_t)rd16(v); return true;
                case 'S': o = (float)rd16(v); return true;
                case 'i': o = (float)(int32_t)rd32(v); return true;
                case 'I': o = (float)rd32(v); return true;
                case 'f': { uint32_t u = rd32(v); float f; std::memcpy(&f, &u, 4); o = f; return true; }
                default: return false;
            }
        };
        const bool tags_ok = walk_tags(tags, end, [&](uint8_t t0, uint8_t t1, uint8_t typ, uint8_t sub, int64_t count, const uint8_t* val,
                                                     const uint8_t*, size_t) {
            const bool arr = typ == 'B' && sub == 'C' && count == (int64_t)l_seq;
            if (t0 == 'f' && t1 == 'i') { if (!tfi && arr) tfi = val; }
            else if (t0 == 'r' && t1 == 'i') { if (!tri && arr) tri = val; }
            else if (t0 == 'f' && t1 == 'p') { if (!tfp && arr) tfp = val; }
            else if (t0 == 'r' && t1 == 'p') { if (!trp && arr) trp = val; }
            else if (t0 == 'f' && t1 == 'n') { if (!has_fn) has_fn = as_num(typ, val, fn); }
            else if (t0 == 'r' && t1 == 'n') { if (!has_rn) has_rn = as_num(typ, val, rn); }
        });
        if (!tags_ok) { delete ob; return fail("corrupt BAM record (auxiliary data)"); }
        const bool usable = l_seq > 0 && tfi && tri && tfp && trp;
        ob->records.insert(ob->records.end(), rec, rec + 4 + bs);
        ob->rec_offset.push_back((int64_t)ob->records.size());
        ob->name_hash.push_back(fnv1a64(body + 32, l_name > 0 ? (size_t)l_name - 1 : 0));
        ob->flag.push_back((int32_t)flag);
        ob->offset.push_back((int64_t)ob->seq.size());
        ob->fn.push_back(has_fn && has_rn ? fn : 0.f);     // extract_features.py:115-119: a KeyError on either tag zeroes both
        ob->rn.push_back(has_fn && has_rn ? rn : 0.f);
        int32_t nsites = 0;
        if (usable) {
            const size_t o = ob->seq.size(), L = l_seq;
            ob->seq.resize(o + L);
            uint8_t* s = ob->seq.data() + o;
            if (flag & 16) {
                for (size_t i = 0; i < L; ++i) {
                    const size_t j = L - 1 - i;
                    s[i] = comp_fwd((uint8_t)kSeqDecode[(packed[j >> 1] >> ((j & 1) ? 0 : 4)) & 15]);
                }
            } else {
                for (size_t i = 0; i < L; ++i) s[i] = (uint8_t)kSeqDecode[(packed[i >> 1] >> ((i & 1) ? 0 : 4)) & 15];
            }
            ob->fi.insert(ob->fi.end(), tfi, tfi + L);
            ob->ri.insert(ob->ri.end(), tri, tri + L);
            ob->fp.insert(ob->fp.end(), tfp, tfp + L);
            ob->rp.insert(ob->rp.end(), trp, trp + L);
            const long n_ = (long)L;
            for (long i = 0; i + 1 < n_; ++i)
                if (s[i] == 'C' && s[i + 1] == 'G') {
                    const long rl = n_ - 2 - i;
                    nsites += (i >= 10 && i < n_ - 10 && rl >= 10 && rl < n_ - 10) ? 1 : 0;
                }
            ob->length.push_back((int32_t)L);
        } else {
            ob->length.push_back(0);
        }
        ob->n_sites.push_back(nsites);
        r->pos += 4 + bs;
    }
    const int32_t nr = (int32_t)ob->flag.size();
    if (nr == 0) { delete ob; return 0; }
    ob->view.n_reads = nr;
    ob->view.records = ob->records.data();
    ob->view.rec_offset = ob->rec_offset.data();
    ob->view.flag = ob->flag.data();
    ob->view.offset = ob->offset.data();
    ob->view.length = ob->length.data();
    ob->view.n_sites = ob->n_sites.data();
    ob->view.seq = ob->seq.data();
    ob->view.fi = ob->fi.data();
    ob->view.ri = ob->ri.data();
    ob->view.fp = ob->fp.data();
    ob->view.rp = ob->rp.data();
    ob->view.fn = ob->fn.data();
    ob->view.rn = ob->rn.data();
    ob->view.name_hash = ob->name_hash.data();
    ob->view.total_bases = (int64_t)ob->seq.size();
    ob->view.voffset_end = r->voffset(r->abs_pos());
    *out = &ob->view;
    return 0;
} CCSM_BAM_CATCH

namespace {
// continue reading at a BGZF virtual offset; limit = last block file offset that may be read (~0: none), stop = chunk end (~0: none)
int reposition(ccsm_bam_reader* r, uint64_t voffset_start, uint64_t limit, uint64_t stop) {
    const uint64_t coff = voffset_start >> 16, uoff = voffset_start & 0xffffu;
    if (fseeko(r->fh, (off_t)coff, SEEK_SET) != 0) return fail("seek failed");
    r->erased += r->stream.size();          // absolute positions stay monotonic (the virtual-offset bookkeeping keys on them)
    r->stream.clear();
    r->pos = 0;
    r->bpos.clear();
    r->next_coffset = coff;
    r->file_eof = false;
    r->limit_coffset = limit;
    r->stop_coffset = stop;
    if (uoff) {
        if (r->fill((size_t)uoff)) return 1;
        if (r->avail() < (size_t)uoff) return fail("seek beyond the end of the BGZF block");
        r->pos += (size_t)uoff;
    }
    return 0;
}
}  // namespace

int ccsm_bam_seek(ccsm_bam_reader* r, uint64_t voffset_start, uint64_t voffset_end) try {
    if (!r) return fail("reader must be non-NULL");
    return reposition(r, voffset_start, voffset_end == 0 ? ~0ull : (voffset_end >> 16), ~0ull);
} CCSM_BAM_CATCH

namespace {
// Is there a BAM record at p?  `have` bytes are readable there.  -1: no; 0: cannot tell yet (more bytes needed: *need); 1: yes, and it
// ends exactly where its block_size says (fixed fields in range, NUL-terminated name without inner NULs, CIGAR operations < 9, the
// auxiliary fields walk to the very end of the record).  A random offset passes with negligible probability, and the caller's
// hand-over check (a chunk's first record must be where the previous chunk's last one ended) catches what is left.
int record_at(const uint8_t* p, size_t have, int32_t n_ref, size_t* need) {
    if (have < 36) { *need = 36; return 0; }
    const uint32_t bs = rd32(p);
    if (bs < 32 + 2 || bs > (1u << 29)) return -1;
    const uint8_t* body = p + 4;
    const int32_t tid = (int32_t)rd32(body), pos = (int32_t)rd32(body + 4);
    const uint32_t l_name = body[8], n_cig = rd16(body + 12), l_seq = rd32(body + 16);
    const int32_t mtid = (int32_t)rd32(body + 20), mpos = (int32_t)rd32(body + 24);
    if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1 || l_name < 1 || l_seq > (1u << 29)) return -1;
    const size_t fixed = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (fixed > bs) return -1;
    if (have < 36 + (size_t)l_name) { *need = 36 + (size_t)l_name; return 0; }
    if (body[32 + l_name - 1] != 0 || std::memchr(body + 32, 0, l_name - 1) != nullptr) return -1;
    for (uint32_t i = 0; i + 1 < l_name; ++i) if (body[32 + i] < 0x21 || body[32 + i] > 0x7e) return -1;
    if (have < 4 + (size_t)bs) { *need = 4 + (size_t)bs; return 0; }
    const uint8_t* cig = body + 32 + l_name;
    for (uint32_t c = 0; c < n_cig; ++c) if ((rd32(cig + 4 * c) & 15) > 8) return -1;
    bool names_ok = true;
    const bool walked = walk_tags(body + fixed, body + bs, [&](uint8_t t0, uint8_t t1, uint8_t, uint8_t, int64_t, const uint8_t*, const uint8_t*, size_t) {
        const bool a0 = (t0 >= 'A' && t0 <= 'Z') || (t0 >= 'a' && t0 <= 'z');
        const bool a1 = a0 && ((t1 >= 'A' && t1 <= 'Z') || (t1 >= 'a' && t1 <= 'z') || (t1 >= '0' && t1 <= '9'));
        names_ok = names_ok && a1;
    });
    return walked && names_ok ? 1 : -1;
}
}  // namespace

int ccsm_bam_seek_chunk(ccsm_bam_reader* r, uint64_t coffset_lo, uint64_t coffset_hi, uint64_t* voffset_first) try {
    if (!r || !voffset_first) return fail("reader and voffset_first must be non-NULL");
    if (coffset_hi <= coffset_lo) return fail("empty chunk");
    *voffset_first = 0;
    // 1. the first BGZF block at or behind coffset_lo: a gzip member header with the 'BC' subfield whose size leads to another one (or
    //    to the end of the file), three deep
    if (fseeko(r->fh, 0, SEEK_END) != 0) return fail("seek failed");
    const uint64_t fsize = (uint64_t)ftello(r->fh);
    auto block_at = [&](uint64_t off, uint64_t* next) -> bool {      // header check only
        uint8_t h[18];
        if (off + 18 > fsize) return false;
        if (fseeko(r->fh, (off_t)off, SEEK_SET) != 0 || std::fread(h, 1, 18, r->fh) != 18) return false;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return false;
        const uint32_t xlen = rd16(h + 10);
        if (xlen < 6) return false;
        uint32_t bsize = 0;
        if (xlen == 6) {
            if (h[12] != 66 || h[13] != 67 || rd16(h + 14) != 2) return false;
            bsize = rd16(h + 16);
        } else {                                                      // other extra subfields: walk them
            std::vector<uint8_t> x(xlen);
            std::memcpy(x.data(), h + 12, 6);
            if (std::fread(x.data() + 6, 1, xlen - 6, r->fh) != xlen - 6) return false;
            bool found = false;
            for (size_t o = 0; o + 4 <= xlen;) {
                const uint32_t sl = rd16(x.data() + o + 2);
                if (x[o] == 66 && x[o + 1] == 67 && sl == 2 && o + 6 <= xlen) { bsize = rd16(x.data() + o + 4); found = true; }
                o += 4 + sl;
            }
            if (!found) return false;
        }
        if ((uint64_t)bsize + 1 < (uint64_t)xlen + 20 || off + bsize + 1 > fsize) return false;
        *next = off + bsize + 1;
        return true;
    };
    auto chain_ok = [&](uint64_t off) -> bool {
        for (int d = 0; d < 3; ++d) {
            if (off == fsize) return true;
            uint64_t nx;
            if (!block_at(off, &nx)) return false;
            off = nx;
        }
        return true;
    };
    uint64_t blk = ~0ull;
    if (coffset_lo >= fsize) return 0;                                  // behind the file: an empty chunk
    if (coffset_lo == 0) {                                              // the first chunk begins behind the header, no search
        if ((r->first_voffset >> 16) >= coffset_hi) return 0;
        if (reposition(r, r->first_voffset, ~0ull, coffset_hi)) return 1;
        if (r->fill(1)) return 1;
        if (r->avail() == 0) return 0;                                  // a file without records
        *voffset_first = r->voffset(r->abs_pos());
        return 0;
    }
    {
        // a block is at most 64 KiB, so one lies in any 64 KiB window that is not the tail of the last block
        std::vector<uint8_t> win((size_t)std::min<uint64_t>(fsize - coffset_lo, 65536 + 4));
        if (fseeko(r->fh, (off_t)coffset_lo, SEEK_SET) != 0 || std::fread(win.data(), 1, win.size(), r->fh) != win.size()) return fail("read failed");
        for (size_t i = 0; i + 4 <= win.size(); ++i)
            if (win[i] == 0x1f && win[i + 1] == 0x8b && win[i + 2] == 8 && win[i + 3] == 4 && chain_ok(coffset_lo + i)) { blk = coffset_lo + i; break; }
    }
    if (blk == ~0ull || blk >= coffset_hi) return 0;                    // no block starts in [lo, hi)
    // 2. the first record that starts in a block of [blk, hi): inflate from blk and test every offset
    if (fseeko(r->fh, (off_t)blk, SEEK_SET) != 0) return fail("seek failed");
    r->erased += r->stream.size();
    r->stream.clear();
    r->pos = 0;
    r->bpos.clear();
    r->next_coffset = blk;
    r->file_eof = false;
    r->limit_coffset = ~0ull;
    r->stop_coffset = coffset_hi;
    size_t p = 0, want = 36;
    for (;;) {
        if (r->fill(p + want)) return 1;                              // pos stays 0 while we search: fill counts from it
        const size_t have = r->stream.size() > p ? r->stream.size() - p : 0;
        if (have < 4) return 0;                                        // ran off the end of the file: no record starts in the chunk
        if ((r->voffset(r->erased + p) >> 16) >= coffset_hi) return 0;  // the search left the chunk: no record starts in it
        size_t need = 0;
        int v = record_at(r->stream.data() + p, have, r->n_ref, &need);
        if (v == 0) {
            if (have >= need) v = -1;                                  // cannot happen; be safe
            else if (r->file_eof) v = -1;                              // would run over the end of the file
            else { want = need; continue; }
        }
        if (v == 1) {                                                  // and the record behind it must be one too (or the file ends)
            const size_t q = p + 4 + rd32(r->stream.data() + p);
            size_t nneed = 36;
            for (;;) {
                if (r->fill(q + nneed)) return 1;
                const size_t h2 = r->stream.size() > q ? r->stream.size() - q : 0;
                if (h2 == 0 && r->file_eof) break;                     // the candidate is the last record of the file
                size_t n2 = 0;
                const int v2 = record_at(r->stream.data() + q, h2, r->n_ref, &n2);
                if (v2 == 0 && !r->file_eof && h2 < n2) { nneed = n2; continue; }
                if (v2 != 1) v = -1;
                break;
            }
        }
        if (v == 1) break;
        ++p;
        want = 36;
    }
    r->pos = p;
    *voffset_first = r->voffset(r->abs_pos());
    return 0;
} CCSM_BAM_CATCH

// Virtual offset a reader reports behind the LAST record of the file: (file offset of the last non-empty BGZF block << 16) | its ISIZE.
// Block headers and trailers only (nothing is inflated); the reader's position is left alone (own file handle).  A multi-GPU run closes
// its hand-over chain on this value: a chain that ends anywhere else has lost the tail of the file (a truncated last record, a chunk whose
// search ran off the end), which the sequential reader reports as an error and the chunked one must not pass over in silence.
int ccsm_bam_eof_voffset(ccsm_bam_reader* r, uint64_t* voffset) try {
    if (!r || !voffset) return fail("arguments must be non-NULL");
    *voffset = 0;
    FILE* fh = std::fopen(r->path.c_str(), "rb");
    if (!fh) return fail("cannot reopen " + r->path);
    struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{fh};
    if (fseeko(fh, 0, SEEK_END) != 0) return fail("seek failed");
    const uint64_t fsize = (uint64_t)ftello(fh);
    // one block header + trailer: -> 1 ok (next, isize), 0 not a block, sets `why`
    std::string why;
    auto block_at = [&](uint64_t off, uint64_t* next, uint32_t* isize) -> bool {
        uint8_t h[12];
        if (off + 12 > fsize || fseeko(fh, (off_t)off, SEEK_SET) != 0 || std::fread(h, 1, 12, fh) != 12) { why = "truncated BGZF block"; return false; }
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) { why = "not a BGZF stream (bad gzip member header)"; return false; }
        const uint32_t xlen = rd16(h + 10);
        std::vector<uint8_t> x(xlen);
        if (std::fread(x.data(), 1, xlen, fh) != xlen) { why = "truncated BGZF block"; return false; }
        int bsize = -1;
        for (size_t o = 0; o + 4 <= xlen;) {
            const uint32_t sl = rd16(x.data() + o + 2);
            if (x[o] == 66 && x[o + 1] == 67 && sl == 2 && o + 6 <= xlen) bsize = rd16(x.data() + o + 4);
            o += 4 + sl;
        }
        if (bsize < 0) { why = "gzip member without the BGZF 'BC' field"; return false; }
        if ((uint64_t)bsize + 1 < (uint64_t)xlen + 20 || off + (uint64_t)bsize + 1 > fsize) { why = "truncated BGZF block"; return false; }
        uint8_t t[4];
        if (fseeko(fh, (off_t)(off + (uint64_t)bsize + 1 - 4), SEEK_SET) != 0 || std::fread(t, 1, 4, fh) != 4) { why = "truncated BGZF block"; return false; }
        *isize = rd32(t);
        if (*isize > 65536) { why = "corrupt BGZF block (ISIZE > 64 KiB)"; return false; }
        *next = off + (uint64_t)bsize + 1;
        return true;
    };
    // walk from `start` to the end of the file: -> 1 reached the end exactly, 0 not a chain of blocks
    auto walk = [&](uint64_t start, uint64_t* last, uint32_t* last_isize) -> bool {
        *last = ~0ull;
        for (uint64_t off = start; off < fsize;) {
            uint64_t nx; uint32_t isz;
            if (!block_at(off, &nx, &isz)) return false;
            if (isz) { *last = off; *last_isize = isz; }
            off = nx;
        }
        return true;
    };
    uint64_t last = ~0ull; uint32_t last_isize = 0;
    // the tail of a large file first: a candidate header in a 64 KiB window (a BGZF block is at most 64 KiB: a window inside the data holds
    // at least one true header) from which the block chain reaches the end of the file with data in it.  The window steps back - 256 KiB,
    // 4 MiB, 64 MiB before the end - before the walk from the first block is paid (one seek + two reads per block: millions of system
    // calls on a 100 GiB input): only a tail of more than 64 MiB of EMPTY blocks gets there.
    for (uint64_t back = 1u << 18; last == ~0ull && back <= (1u << 26) && fsize > back; back <<= 4) {
        const uint64_t base = fsize - back;
        std::vector<uint8_t> win(65536 + 4);
        if (fseeko(fh, (off_t)base, SEEK_SET) == 0 && std::fread(win.data(), 1, win.size(), fh) == win.size()) {
            for (size_t i = 0; i + 4 <= win.size(); ++i) {
                if (win[i] != 0x1f || win[i + 1] != 0x8b || win[i + 2] != 8 || win[i + 3] != 4) continue;
                if (walk(base + i, &last, &last_isize) && last != ~0ull) break;
                last = ~0ull;
            }
        }
    }
    if (last == ~0ull) {                                             // small file, or a tail of empty blocks: from the first block
        why.clear();
        if (!walk(0, &last, &last_isize)) return fail(why.empty() ? "not a BGZF stream" : why);
    }
    if (last == ~0ull) return fail("no data in the BGZF stream");
    *voffset = (last << 16) | last_isize;
    return 0;
} CCSM_BAM_CATCH

int ccsm_bam_tell(ccsm_bam_reader* r, uint64_t* voffset) {
    if (!r || !voffset) return fail("arguments must be non-NULL");
    if (r->fill(1)) return 1;
    *voffset = r->voffset(r->abs_pos());
    return 0;
}

int64_t ccsm_bam_inflated_bytes(const ccsm_bam_reader* r) { return r ? (int64_t)r->inflated : 0; }

int ccsm_bam_writer_open(const char* path, const char* header_text, int64_t text_len, const uint8_t* refs, int64_t refs_len,
                         int32_t n_ref, int threads, int level, ccsm_bam_writer** out) try {
    if (!path || !out || (text_len > 0 && !header_text) || (refs_len > 0 && !refs)) return fail("bad arguments");
    *out = nullptr;
    if (level < 1 || level > 9) return fail("level must be in [1, 9]");
    ccsm_bam_writer* w = new (std::nothrow) ccsm_bam_writer();
    if (!w) return fail("out of memory");
    w->threads = std::max(1, threads);
    w->level = level;
    w->fh = std::fopen(path, "wb");
    if (!w->fh) { delete w; return fail(std::string("cannot create ") + path); }
    uint8_t h[8] = {'B', 'A', 'M', 1};
    wr32(h + 4, (uint32_t)text_len);
    w->put(h, 8);
    if (text_len > 0) w->put(reinterpret_cast<const uint8_t*>(header_text), (size_t)text_len);
    uint8_t nr[4];
    wr32(nr, (uint32_t)n_ref);
    w->put(nr, 4);
    if (refs_len > 0) w->put(refs, (size_t)refs_len);
    *out = w;
    return 0;
} CCSM_BAM_CATCH

int ccsm_bam_write_batch(ccsm_bam_writer* w, const ccsm_bam_batch* b, const int32_t* first_site, const int32_t* locs,
                         const float* prob1, const uint8_t* tagged, int rm_pulse, int32_t* n_tagged) try {
    if (!w || !b) return fail("writer and batch must be non-NULL");
    int32_t cnt = 0;
    std::vector<uint8_t> rec;
    std::string mm;
    std::vector<uint8_t> ml;
    for (int32_t r = 0; r < b->n_reads; ++r) {
        const uint8_t* src = b->records + b->rec_offset[r];
        const uint32_t bs = rd32(src);
        const uint8_t* body = src + 4;
        const uint32_t l_name = body[8], n_cig = rd16(body + 12), l_seq = rd32(body + 16);
        const size_t fixed = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
        rec.assign(src, src + 4 + fixed);
        walk_tags(body + fixed, body + bs, [&](uint8_t t0, uint8_t t1, uint8_t, uint8_t, int64_t, const uint8_t*, const uint8_t* whole, size_t len) {
            if (t0 == 'M' && (t1 == 'M' || t1 == 'L')) return;
            if (rm_pulse && ((t0 == 'f' || t0 == 'r') && (t1 == 'i' || t1 == 'p'))) return;
            rec.insert(rec.end(), whole, whole + len);
        });
        bool add = tagged && tagged[r] && first_site && locs && prob1 && b->length[r] > 0 && first_site[r + 1] > first_site[r];
        if (add) {
            // _convert_locs_to_mmtag: one forward scan over the C's of the forward sequence; a location that is not the next C
            // at or after the previous match leaves the read untagged
            const uint8_t* s = b->seq + b->offset[r];
            const int32_t L = b->length[r];
            mm.assign("C+m?");
            ml.clear();
            int32_t p = 0, order = -1, prev_order = -1;
            for (int32_t k = first_site[r]; k < first_site[r + 1] && add; ++k) {
                const int32_t loc = locs[k];
                if (loc < p || loc >= L || s[loc] != 'C') { add = false; break; }
                for (; p < loc; ++p) order += (s[p] == 'C') ? 1 : 0;
                order += 1;              // the C at loc itself
                p = loc + 1;
                mm += ',';
                mm += std::to_string(prev_order < 0 ? order : order - 1 - prev_order);
                prev_order = order;
                const float pr = prob1[k];
                ml.push_back(pr < 1.0f ? (uint8_t)std::floor(pr * 256.0f) : (uint8_t)255);
            }
            mm += ';';
        }
        if (add) {
            rec.push_back('M'); rec.push_back('M'); rec.push_back('Z');
            rec.insert(rec.end(), mm.begin(), mm.end());
            rec.push_back(0);
            rec.push_back('M'); rec.push_back('L'); rec.push_back('B'); rec.push_back('C');
            uint8_t c4[4];
            wr32(c4, (uint32_t)ml.size());
            rec.insert(rec.end(), c4, c4 + 4);
            rec.insert(rec.end(), ml.begin(), ml.end());
            ++cnt;
        }
        wr32(rec.data(), (uint32_t)(rec.size() - 4));
        const uint64_t ubeg = w->run_upos;
        if (w->put(rec.data(), rec.size())) return 1;
        if (w->track) w->note_record(rec.data() + 4, ubeg, w->run_upos);
    }
    if (n_tagged) *n_tagged = cnt;
    return 0;
} CCSM_BAM_CATCH

int ccsm_bam_writer_flush(ccsm_bam_writer* w, int64_t* file_offset) try {
    if (!w) return fail("writer must be non-NULL");
    if (w->flush_blocks(true)) return 1;
    if (std::fflush(w->fh) != 0) return fail("write failed");
    w->end_run();
    if (file_offset) *file_offset = (int64_t)w->file_off;
    return 0;
} CCSM_BAM_CATCH

int ccsm_bam_writer_track_index(ccsm_bam_writer* w, int enable) {
    if (!w) return fail("writer must be non-NULL");
    if (w->run_upos != 0) return fail("index tracking can only be switched at a run boundary (right after ccsm_bam_writer_flush)");
    w->track = enable != 0;
    return 0;
}

int ccsm_bam_writer_take_index(ccsm_bam_writer* w, ccsm_bam_index_run* out) try {
    if (!w || !out) return fail("writer and out must be non-NULL");
    if (!w->track) return fail("index tracking is off");
    if (w->run_upos != 0) return fail("ccsm_bam_writer_take_index must follow ccsm_bam_writer_flush");
    w->handed.swap(w->done);
    w->done.clear();
    out->n_records = w->acc_records;
    out->n_unplaced = w->acc_unplaced;
    out->sorted = w->acc_sorted ? 1 : 0;
    out->first_k1 = w->acc_first_k1; out->first_k2 = w->acc_first_k2;
    out->last_k1 = w->acc_last_k1; out->last_k2 = w->acc_last_k2;
    out->n_entries = (int64_t)w->handed.size();
    out->entries = w->handed.data();
    out->file_start = w->acc_file_start < 0 ? (int64_t)w->file_off : w->acc_file_start;
    out->file_end = (int64_t)w->file_off;
    w->acc_records = w->acc_unplaced = 0;
    w->acc_sorted = true; w->acc_any = false;
    w->acc_file_start = -1;
    return 0;
} CCSM_BAM_CATCH

int ccsm_bam_writer_close(ccsm_bam_writer* w) try {
    if (!w) return 0;
    int rc = w->flush_blocks(true);
    if (rc == 0 && std::fwrite(kBgzfEof, 1, sizeof(kBgzfEof), w->fh) != sizeof(kBgzfEof)) rc = fail("write failed");
    if (std::fclose(w->fh) != 0 && rc == 0) rc = fail("close failed");
    delete w;
    return rc;
} CCSM_BAM_CATCH

int ccsm_bam_modcalls_of_batch(const ccsm_bam_batch* b, const ccsm_bam_modcall_opts* opts, const uint8_t* const* site_mask,
                               const int64_t* mask_len, int32_t n_ref, int threads, ccsm_bam_modcalls** out) try {
    if (!b || !opts || !out) return fail("batch, opts and out must be non-NULL");
    *out = nullptr;
    if (opts->refsites_all && (!site_mask || !mask_len)) return fail("refsites_all needs the reference site masks");
    const int n = b->n_reads;
    const int parts = std::max(1, std::min(threads, n));
    std::vector<CallRows> rows((size_t)parts);
    parallel_for(parts, parts, [&](int p) {
        std::vector<std::pair<int32_t, uint8_t>> calls;
        const int r0 = (int)((int64_t)n * p / parts), r1 = (int)((int64_t)n * (p + 1) / parts);
        for (int r = r0; r < r1; ++r)
            modcalls_of_record(b->records + b->rec_offset[r], *opts, site_mask, mask_len, n_ref, calls, rows[(size_t)p]);
    });
    OwnedCalls* oc = new (std::nothrow) OwnedCalls();
    if (!oc) return fail("out of memory");
    int64_t used = 0;
    for (auto& cr : rows) {
        oc->tid.insert(oc->tid.end(), cr.tid.begin(), cr.tid.end());
        oc->pos.insert(oc->pos.end(), cr.pos.begin(), cr.pos.end());
        oc->strand.insert(oc->strand.end(), cr.strand.begin(), cr.strand.end());
        oc->ml.insert(oc->ml.end(), cr.ml.begin(), cr.ml.end());
        oc->hap.insert(oc->hap.end(), cr.hap.begin(), cr.hap.end());
        used += cr.used;
    }
    oc->view.n = (int64_t)oc->tid.size();
    oc->view.tid = oc->tid.data();
    oc->view.pos = oc->pos.data();
    oc->view.strand = oc->strand.data();
    oc->view.ml = oc->ml.data();
    oc->view.hap = oc->hap.data();
    oc->view.n_records = n;
    oc->view.n_used = used;
    *out = &oc->view;
    return 0;
} CCSM_BAM_CATCH

void ccsm_bam_modcalls_free(ccsm_bam_modcalls* c) { delete reinterpret_cast<OwnedCalls*>(c); }

// ---- BAI index and coordinate sort (the reference's pysam.sort + pysam.index post-processing, call_modifications.py:592-607) ----
namespace {

inline int reg2bin(int64_t beg, int64_t end) {      // SAM spec 5.3, 14-bit minimum interval, 5 levels
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

// samtools sort's coordinate order: reference id with unmapped (-1) last, position, forward strand before reverse
inline void sort_key(const uint8_t* body, uint64_t& k1, uint32_t& k2) {
    const uint32_t tid = rd32(body);
    const int32_t pos = (int32_t)rd32(body + 4);
    k1 = ((uint64_t)tid << 32) | (uint32_t)(pos + 1);
    k2 = (rd16(body + 14) & 16) ? 1 : 0;
}

struct RefIndex {
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
    uint64_t off_beg = ~0ull, off_end = 0, n_mapped = 0, n_unmapped = 0;
};

// BAI index (SAM spec 5.2) fed record by record in file order: binning index, 16 kb linear index, htslib's metadata pseudo-bin
// 37450, n_no_coor; plus the check that the records are in samtools' coordinate order.
struct BaiBuilder {
    std::vector<RefIndex> refs;
    uint64_t n_no_coor = 0, prev1 = 0;
    uint32_t prev2 = 0;
    bool is_sorted = true, first = true;
    int64_t count = 0;
    int save_bin = -1, save_tid = -2;
    uint64_t save_off = 0, last_off = 0;
    explicit BaiBuilder(int32_t n_ref) : refs((size_t)std::max(0, n_ref)) {}
    void flush_bin() {
        if (save_bin >= 0 && save_tid >= 0 && save_tid < (int)refs.size()) refs[(size_t)save_tid].bins[(uint32_t)save_bin].emplace_back(save_off, last_off);
    }
    void order(uint64_t k1, uint32_t k2) {
        if (!first && (k1 < prev1 || (k1 == prev1 && k2 < prev2))) is_sorted = false;
        first = false; prev1 = k1; prev2 = k2;
    }
    // [b, e) = the record's span on the reference as index_span() gives it; returns 1 (with the error set) on a record a BAI cannot hold
    int add(int32_t tid, int64_t b, int64_t e, bool unmapped, uint64_t beg_off, uint64_t end_off) {
        ++count;
        if (tid >= (int32_t)refs.size()) return fail("record refers to a reference id beyond the header");
        if (tid < 0) {
            ++n_no_coor;
            if (save_tid >= 0) { last_off = beg_off; flush_bin(); save_bin = -1; save_tid = -1; }
            last_off = end_off;
            return 0;
        }
        RefIndex& ri = refs[(size_t)tid];
        if (b >= (1LL << 29)) return fail("record position beyond the range of a BAI index (2^29)");
        if (e > (1LL << 29)) e = 1LL << 29;
        const int bin = reg2bin(b, e);
        if (bin != save_bin || tid != save_tid) {
            last_off = beg_off;
            flush_bin();
            save_bin = bin; save_tid = tid; save_off = beg_off;
        }
        if (!unmapped) {
            const size_t w0 = (size_t)(b >> 14), w1 = (size_t)((e - 1) >> 14);
            if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, ~0ull);
            for (size_t w = w0; w <= w1; ++w) if (ri.linear[w] == ~0ull) ri.linear[w] = beg_off;
            ++ri.n_mapped;
        } else {
            ++ri.n_unmapped;
        }
        if (ri.off_beg == ~0ull) ri.off_beg = beg_off;
        ri.off_end = end_off;
        last_off = end_off;
        return 0;
    }
    // unplaced records that are not fed one by one (the writer's run tables only count them)
    void add_unplaced(int64_t n) {
        if (n <= 0) return;
        if (save_tid >= 0) { flush_bin(); save_bin = -1; save_tid = -1; }
        n_no_coor += (uint64_t)n;
        count += n;
    }
    int write(const char* bai_path) {
        flush_bin();
        save_bin = -1;
        FILE* f = std::fopen(bai_path, "wb");
        if (!f) return fail(std::string("cannot write ") + bai_path);
        std::vector<uint8_t> out;
        auto p32 = [&](uint32_t v) { uint8_t b[4]; wr32(b, v); out.insert(out.end(), b, b + 4); };
        auto p64 = [&](uint64_t v) { p32((uint32_t)v); p32((uint32_t)(v >> 32)); };
        out.insert(out.end(), {'B', 'A', 'I', 1});
        p32((uint32_t)refs.size());
        for (auto& ri : refs) {
            const bool any = ri.off_beg != ~0ull;
            p32((uint32_t)(ri.bins.size() + (any ? 1 : 0)));
            for (auto& kv : ri.bins) {
                p32(kv.first);
                p32((uint32_t)kv.second.size());
                for (auto& ch : kv.second) { p64(ch.first); p64(ch.second); }
            }
            if (any) {                                   // htslib's metadata pseudo-bin
                p32(37450); p32(2);
                p64(ri.off_beg); p64(ri.off_end); p64(ri.n_mapped); p64(ri.n_unmapped);
            }
            for (size_t w = ri.linear.size(); w-- > 0;)  // empty windows take the next window's offset
                if (ri.linear[w] == ~0ull) ri.linear[w] = (w + 1 < ri.linear.size()) ? ri.linear[w + 1] : 0;
            p32((uint32_t)ri.linear.size());
            for (uint64_t v : ri.linear) p64(v);
        }
        p64(n_no_coor);
        const bool okw = std::fwrite(out.data(), 1, out.size(), f) == out.size();
        if (std::fclose(f) != 0 || !okw) return fail("write failed");
        return 0;
    }
};

// [b, e) a record covers on its reference for the index: pos .. pos + reference length of the CIGAR (one base when unmapped / no CIGAR)
inline void index_span(const uint8_t* body, int64_t& b, int64_t& e) {
    const int32_t pos = (int32_t)rd32(body + 4);
    const uint32_t l_name = body[8], n_cig = rd16(body + 12), flag = rd16(body + 14);
    int64_t reflen = 0;
    const uint8_t* cig = body + 32 + l_name;
    for (uint32_t c = 0; c < n_cig; ++c) {
        const uint32_t v = rd32(cig + 4 * c), op = v & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += v >> 4;
    }
    b = pos < 0 ? 0 : pos;
    e = (flag & 4) || reflen == 0 ? b + 1 : b + reflen;
}

}  // namespace

void ccsm_bam_writer::note_record(const uint8_t* body, uint64_t ubeg, uint64_t uend) {
    uint64_t k1; uint32_t k2;
    sort_key(body, k1, k2);
    if (!acc_any) { acc_first_k1 = k1; acc_first_k2 = k2; }
    else if (k1 < acc_last_k1 || (k1 == acc_last_k1 && k2 < acc_last_k2)) acc_sorted = false;
    acc_any = true; acc_last_k1 = k1; acc_last_k2 = k2;
    ++acc_records;
    const int32_t tid = (int32_t)rd32(body);
    if (tid < 0) { ++acc_unplaced; return; }
    int64_t b, e;
    index_span(body, b, e);
    pend.push_back({tid, (int32_t)std::min<int64_t>(b, INT32_MAX), (int32_t)std::min<int64_t>(e, INT32_MAX), rd16(body + 14), ubeg, uend});
}


int ccsm_bam_index_build(const char* bam_path, const char* bai_path, int threads, int* sorted, int64_t* n_records) try {
    if (!bam_path || !bai_path) return fail("paths must be non-NULL");
    ccsm_bam_reader* r = nullptr;
    if (ccsm_bam_open(bam_path, threads, &r)) return 1;
    BaiBuilder bb(r->n_ref);
    int rc = 0;
    for (;;) {
        if (r->fill(4)) { rc = 1; break; }
        if (r->avail() == 0) break;
        if (r->avail() < 4) { rc = fail("truncated BAM record"); break; }
        const uint32_t bs = rd32(r->stream.data() + r->pos);
        if (bs < 32) { rc = fail("corrupt BAM record (block_size < 32)"); break; }
        if (r->fill(4 + (size_t)bs)) { rc = 1; break; }
        if (r->avail() < 4 + (size_t)bs) { rc = fail("truncated BAM record"); break; }
        const uint64_t beg_off = r->voffset(r->abs_pos());
        const uint8_t* body = r->stream.data() + r->pos + 4;
        const int32_t tid = (int32_t)rd32(body);
        const uint32_t l_name = body[8], n_cig = rd16(body + 12), flag = rd16(body + 14);
        if (32 + (size_t)l_name + 4 * (size_t)n_cig > bs) { rc = fail("corrupt BAM record (field lengths exceed block_size)"); break; }
        uint64_t k1; uint32_t k2;
        sort_key(body, k1, k2);
        bb.order(k1, k2);
        int64_t b, e;
        index_span(body, b, e);
        r->pos += 4 + bs;
        const uint64_t end_off = r->voffset(r->abs_pos());
        if ((rc = bb.add(tid, b, e, (flag & 4) != 0, beg_off, end_off)) != 0) break;
    }
    ccsm_bam_close(r);
    if (rc) return rc;
    if (sorted) *sorted = bb.is_sorted ? 1 : 0;
    if (n_records) *n_records = bb.count;
    if (!bb.is_sorted) return 0;                       // an index over unsorted records is meaningless: none is written
    return bb.write(bai_path);
} CCSM_BAM_CATCH

int ccsm_bam_index_write(const char* bai_path, int32_t n_ref, int32_t n_runs, const ccsm_bam_index_run* runs, const int64_t* shift,
                         int* sorted, int64_t* n_records) try {
    if (!bai_path || n_runs < 0 || (n_runs > 0 && !runs)) return fail("bad arguments");
    BaiBuilder bb(n_ref);
    for (int32_t i = 0; i < n_runs; ++i) {
        const ccsm_bam_index_run& r = runs[i];
        if (r.n_records == 0) continue;
        if (!r.sorted) bb.is_sorted = false;
        bb.order(r.first_k1, r.first_k2);
        bb.order(r.last_k1, r.last_k2);
        if (!bb.is_sorted) break;
        const uint64_t sh = (uint64_t)(shift ? shift[i] : 0) << 16;      // two's complement: a negative shift subtracts
        if (r.n_entries > 0 && !r.entries) return fail("run without its entries");
        for (int64_t k = 0; k < r.n_entries; ++k) {
            const ccsm_bam_index_entry& q = r.entries[k];
            if (bb.add(q.tid, q.pos, q.end, (q.flag & 4) != 0, q.vbeg + sh, q.vend + sh)) return 1;
        }
        // in a sorted file a run's unplaced records follow its placed ones
        bb.add_unplaced(r.n_unplaced);
    }
    if (sorted) *sorted = bb.is_sorted ? 1 : 0;
    if (n_records) {
        int64_t n = 0;
        for (int32_t i = 0; i < n_runs; ++i) n += runs[i].n_records;
        *n_records = n;
    }
    if (!bb.is_sorted) return 0;
    return bb.write(bai_path);
} CCSM_BAM_CATCH



// Coordinate sort (samtools sort order: reference id with unplaced reads last, position, reverse-strand flag; stable).  Records are
// collected up to max_bytes (0 = no limit); when the input is larger, each full collection is sorted and spilled as a run
// (<out_path>.sorttmp.<k>.bam, BGZF level 1) and the runs are merged k-way at the end, like samtools sort's temporary files — host
// memory stays bounded by max_bytes plus one decode window per run.
namespace {
struct SortChunk {
    std::vector<uint8_t> data;
    std::vector<uint64_t> off, k1;
    std::vector<uint32_t> k2;
    void clear() { data.clear(); off.clear(); k1.clear(); k2.clear(); }
    // records in sorted order to w
    int write_sorted(ccsm_bam_writer* w) {
        const size_t n = k1.size();
        if (n > 0xfffffffeull) return fail("too many records in one sort run");
        std::vector<uint32_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return k1[x] < k1[y] || (k1[x] == k1[y] && k2[x] < k2[y]); });
        off.push_back(data.size());
        for (size_t i = 0; i < n; ++i) {
            const uint32_t j = order[i];
            if (w->put(data.data() + off[j], (size_t)(off[j + 1] - off[j]))) return 1;
        }
        off.pop_back();
        return 0;
    }
};
// one record at a time from a run file
struct RunCursor {
    ccsm_bam_reader* r = nullptr;
    uint64_t k1 = 0;
    uint32_t k2 = 0, bs = 0;
    bool done = false;
    int advance(bool consume) {             // 0 ok (or done set), 1 error
        if (consume) r->pos += 4 + (size_t)bs;
        if (r->fill(4)) return 1;
        if (r->avail() == 0) { done = true; return 0; }
        if (r->avail() < 4) return fail("truncated sort run");
        bs = rd32(r->stream.data() + r->pos);
        if (bs < 32) return fail("corrupt sort run");
        if (r->fill(4 + (size_t)bs)) return 1;
        if (r->avail() < 4 + (size_t)bs) return fail("truncated sort run");
        sort_key(r->stream.data() + r->pos + 4, k1, k2);
        return 0;
    }
    const uint8_t* rec() const { return r->stream.data() + r->pos; }
};
}  // namespace

int ccsm_bam_sort(const char* in_path, const char* out_path, int threads, int level, int64_t max_bytes) try {
    if (!in_path || !out_path) return fail("paths must be non-NULL");
    ccsm_bam_reader* r = nullptr;
    if (ccsm_bam_open(in_path, threads, &r)) return 1;
    // header: @HD ... SO:coordinate as samtools sort leaves it
    std::string text = r->text;
    {
        std::string hd = "@HD\tVN:1.6\tSO:coordinate\n";
        if (text.compare(0, 3, "@HD") == 0) {
            const size_t eol = text.find('\n');
            std::string line = text.substr(0, eol == std::string::npos ? text.size() : eol);
            const size_t so = line.find("\tSO:");
            if (so != std::string::npos) {
                size_t e = line.find('\t', so + 1);
                line.replace(so, (e == std::string::npos ? line.size() : e) - so, "\tSO:coordinate");
            } else {
                line += "\tSO:coordinate";
            }
            text = line + "\n" + (eol == std::string::npos ? std::string() : text.substr(eol + 1));
        } else {
            text = hd + text;
        }
    }
    const std::vector<uint8_t> refs = r->refs;
    const int32_t n_ref = r->n_ref;
    std::vector<std::string> runs;
    auto remove_runs = [&]() { for (const std::string& p : runs) std::remove(p.c_str()); };
    SortChunk ch;
    auto spill = [&]() -> int {
        const std::string path = std::string(out_path) + ".sorttmp." + std::to_string(runs.size()) + ".bam";
        ccsm_bam_writer* w = nullptr;
        if (ccsm_bam_writer_open(path.c_str(), text.data(), (int64_t)text.size(), refs.data(), (int64_t)refs.size(), n_ref, threads, 1, &w)) return 1;
        runs.push_back(path);
        if (ch.write_sorted(w)) { ccsm_bam_writer_close(w); return 1; }
        ch.clear();
        return ccsm_bam_writer_close(w);
    };
    int rc = 0;
    for (;;) {
        if (r->fill(4)) { rc = 1; break; }
        if (r->avail() == 0) break;
        if (r->avail() < 4) { rc = fail("truncated BAM record"); break; }
        const uint32_t bs = rd32(r->stream.data() + r->pos);
        if (bs < 32) { rc = fail("corrupt BAM record (block_size < 32)"); break; }
        if (r->fill(4 + (size_t)bs)) { rc = 1; break; }
        if (r->avail() < 4 + (size_t)bs) { rc = fail("truncated BAM record"); break; }
        if (max_bytes > 0 && !ch.k1.empty() && (int64_t)(ch.data.size() + 4 + bs) > max_bytes && (rc = spill()) != 0) break;
        const uint8_t* rec = r->stream.data() + r->pos;
        uint64_t a; uint32_t b;
        sort_key(rec + 4, a, b);
        ch.off.push_back(ch.data.size()); ch.k1.push_back(a); ch.k2.push_back(b);
        ch.data.insert(ch.data.end(), rec, rec + 4 + bs);
        r->pos += 4 + bs;
    }
    ccsm_bam_close(r);
    if (rc == 0 && !runs.empty() && !ch.k1.empty()) rc = spill();      // the last partial collection becomes a run too
    if (rc) { remove_runs(); return rc; }
    ccsm_bam_writer* w = nullptr;
    if (ccsm_bam_writer_open(out_path, text.data(), (int64_t)text.size(), refs.data(), (int64_t)refs.size(), n_ref, threads, level, &w)) { remove_runs(); return 1; }
    if (runs.empty()) {
        if (ch.write_sorted(w)) { ccsm_bam_writer_close(w); return 1; }
        return ccsm_bam_writer_close(w);
    }
    // k-way merge; ties go to the earlier run (= earlier in the input: the sort stays stable)
    std::vector<RunCursor> cur(runs.size());
    for (size_t i = 0; i < runs.size() && rc == 0; ++i) {
        if (ccsm_bam_open(runs[i].c_str(), 1, &cur[i].r)) { rc = 1; break; }
        rc = cur[i].advance(false);
    }
    auto later = [&](size_t x, size_t y) {       // priority_queue keeps the LARGEST on top: order by "comes later"
        if (cur[x].k1 != cur[y].k1) return cur[x].k1 > cur[y].k1;
        if (cur[x].k2 != cur[y].k2) return cur[x].k2 > cur[y].k2;
        return x > y;
    };
    std::priority_queue<size_t, std::vector<size_t>, decltype(later)> heap(later);
    for (size_t i = 0; i < cur.size() && rc == 0; ++i)
        if (!cur[i].done) heap.push(i);
    while (rc == 0 && !heap.empty()) {
        const size_t i = heap.top();
        heap.pop();
        if (w->put(cur[i].rec(), 4 + (size_t)cur[i].bs)) { rc = 1; break; }
        rc = cur[i].advance(true);
        if (rc == 0 && !cur[i].done) heap.push(i);
    }
    for (RunCursor& c : cur)
        if (c.r) ccsm_bam_close(c.r);
    remove_runs();
    const int rc2 = ccsm_bam_writer_close(w);
    return rc ? rc : rc2;
} catch (const std::bad_alloc&) {
    return fail("out of host memory while sorting (lower the sort memory limit: the sort then spills runs to disk)");
} catch (const std::exception& e) {
    return fail(std::string("sort: ") + e.what());
}

int ccsm_bam_align_info(const ccsm_bam_batch* b, int32_t* mapq, int32_t* qstart, int32_t* qend, double* identity) try {
    if (!b || !mapq || !qstart || !qend || !identity) return fail("arguments must be non-NULL");
    for (int32_t r = 0; r < b->n_reads; ++r) {
        const uint8_t* body = b->records + b->rec_offset[r] + 4;
        const uint32_t l_name = body[8], n_cig = rd16(body + 12), l_seq = rd32(body + 16);
        const uint8_t* cig = body + 32 + l_name;
        int64_t cnt[16] = {0};
        for (uint32_t c = 0; c < n_cig; ++c) { const uint32_t v = rd32(cig + 4 * c); cnt[v & 15] += v >> 4; }
        int64_t qs = 0, qe = l_seq;
        for (uint32_t c = 0; c < n_cig; ++c) {                       // leading clips (pysam query_alignment_start)
            const uint32_t v = rd32(cig + 4 * c), op = v & 15;
            if (op == 5) continue;
            if (op == 4) { qs += v >> 4; continue; }
            break;
        }
        for (uint32_t c = n_cig; c-- > 0;) {                          // trailing clips (query_alignment_end)
            const uint32_t v = rd32(cig + 4 * c), op = v & 15;
            if (op == 5) continue;
            if (op == 4) { qe -= v >> 4; continue; }
            break;
        }
        const int64_t nalign = cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[6] + cnt[7] + cnt[8] + cnt[9];
        mapq[r] = body[9];
        qstart[r] = (int32_t)qs;
        qend[r] = (int32_t)qe;
        identity[r] = nalign > 0 ? (double)(cnt[0] + cnt[7]) / (double)nalign : 0.0;
    }
    return 0;
} CCSM_BAM_CATCH

}  // extern "C"
