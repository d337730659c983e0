// fl_jpeghuff_dev.hip -- Huffman decoding of sequential JPEG scans ON THE DEVICE (gfx950), so that the host's share of a JPEG
// request is a header parse and one copy.  Reference: src/handler.rs:205-220 (ImageReader -> JpegDecoder -> DynamicImage::from_decoder,
// zune-jpeg 0.4.14); the code words are ITU-T T.81 Annex F.2.2, as in the host decoder (fl_jpeghuff.cpp), whose coefficients this
// must reproduce bit for bit (tests/test_jpeg_decode.py).
//
// A Huffman stream has no random access, but a decoder started at a wrong bit falls into step with the right one after a few code
// words (the codes are complete prefix codes), and from then on produces the same symbols.  So (Weissenberger & Schmidt's scheme,
// simplified to what fits five small kernels):
//   jh_init     the blob's header, the block words (every block "wide": 64 x i16 at its own fixed place) and zeroed coefficients
//   jh_sync<1>  every subsequence of kJhSubBits bits is decoded from its first bit as if a block started there; the state it ends
//               in -- (bit position, block of the MCU, coefficient index) -- is stored
//   jh_sync<0>  x kJhSyncRounds: every subsequence is decoded again from the stored end state of the one before it; where the end
//               state changes it is stored again.  Subsequence 0 starts at the true start, so after round r the first r states are
//               exact, and in practice (self-synchronisation) all of them are after one or two rounds
//               (every walk also counts the blocks it completes and the sum of DC differences per component; the counts of a
//               subsequence's LAST walk are the ones that stay)
//   jh_scan     checks that the chain of states is consistent (else error bit 1: the host decodes the file instead); exclusive prefix
//               sums of the counts: the number of the block a subsequence starts in, and the DC predictors there
//   jh_write    the walk with values: coefficients go to their block's 64 halfwords, DC terms as prefix + running sum
// All kernels take every picture of a batch at once; a workgroup serves 256 consecutive subsequences of one picture, with the
// picture's four code tables in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "fl_jpegdec.h"

namespace fl {

namespace {

constexpr uint32_t TW = kJhTableWords;
constexpr uint32_t T_LOOK = 1024, T_MAXCODE = 1024 + 256, T_VALOFF = T_MAXCODE + 18, T_VALS = T_VALOFF + 17; // word offsets inside a table

struct Ctx {
    const JpegBlobHeader *H;
    const JpegHuffStage *S;
    const uint32_t *words;  // the unstuffed segment as big-endian words
    uint32_t nwords;
    const uint32_t *lwords; // the workgroup's window of it in LDS: words [lbase, lbase + kWinWords)
    uint32_t lbase;
};
constexpr uint32_t kWinWords = 256u * kJhSubBits / 32u + 8u; // 256 subsequences + the words a walk may read past its end

__device__ __forceinline__ Ctx make_ctx(const JhJob &jb)
{
    Ctx c;
    c.H = reinterpret_cast<const JpegBlobHeader *>(jb.stage);
    c.S = reinterpret_cast<const JpegHuffStage *>(jb.stage + sizeof(JpegBlobHeader));
    c.words = reinterpret_cast<const uint32_t *>(jb.stage + c.S->stream_off);
    c.nwords = (c.S->stream_bits / 8u + 16u) / 4u; // (the stage is padded with 16 bytes of ones)
    c.lwords = nullptr; c.lbase = 0u;
    return c;
}

// The 256 subsequences of a workgroup are one contiguous piece of the segment: it is copied to LDS once (coalesced), and the walks
// read their words there -- a walk consumes a word every five symbols or so, and fetched one by one from the L2 those loads were
// most of its time.
__device__ __forceinline__ void stage_window(Ctx &c, uint32_t first_sub, uint32_t *win)
{
    const uint32_t base = first_sub * (kJhSubBits / 32u);
    for (uint32_t k = threadIdx.x; k < kWinWords; k += blockDim.x) {
        const uint32_t i = base + k;
        win[k] = c.words[i < c.nwords ? i : c.nwords - 1u];
    }
    c.lwords = win; c.lbase = base;
    __syncthreads();
}

__device__ __forceinline__ void stage_tables(const JhJob &jb, const Ctx &c, uint32_t *lut)
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(jb.stage + c.S->tables_off);
    for (uint32_t k = threadIdx.x; k < 4u * TW; k += blockDim.x) lut[k] = src[k];
    __syncthreads();
}

__device__ __forceinline__ uint64_t pack_state(uint32_t p, uint32_t j, uint32_t k) { return (uint64_t)p | ((uint64_t)j << 32) | ((uint64_t)k << 40); }

// One walk over the code words from state (p, j, k) until the bit position reaches p_end.
// MODE 0: states only.  MODE 1: + blocks completed and DC difference sums.  MODE 2: + the coefficients are stored (q0 = number of the
// block the walk starts in, dc0 = the components' DC predictors there) and invalid code words of real blocks are reported.
template <int MODE>
__device__ __forceinline__ uint64_t jh_walk(const Ctx &c, const uint32_t *lut, uint64_t state, uint32_t p_end, int32_t *cnt4, uint32_t q0, const int32_t *dc0,
                                            int16_t *coef, uint32_t *err)
{
    const JpegHuffStage &S = *c.S;
    uint32_t p = (uint32_t)state, j = (uint32_t)(state >> 32) & 15u, k = (uint32_t)(state >> 40) & 127u;
    uint32_t wi = p >> 5;
    auto word = [&](uint32_t i) {
        const uint32_t k = i - c.lbase;
        return __builtin_bswap32(k < kWinWords ? c.lwords[k] : c.words[i < c.nwords ? i : c.nwords - 1u]);
    };
    uint64_t buf = (((uint64_t)word(wi) << 32) | word(wi + 1u)) << (p & 31u);
    int cnt = 64 - (int)(p & 31u);
    wi += 2u;
    uint32_t next_word = word(wi);
    int32_t nblk = 0, dcs[3] = {0, 0, 0};
    uint32_t q = q0, gidx = 0;
    bool bad = false;
    const uint32_t bpm = S.bpm;
    // per block of the MCU, four bits: component (2) ... and eight more: its DC and AC table (2 + 2) -- in registers, not re-read per symbol
    uint64_t comp_of = 0, tabs_of = 0;
    for (uint32_t b = 0; b < bpm; ++b) {
        const uint32_t cb = S.blk_comp[b];
        comp_of |= (uint64_t)cb << (4u * b);
        tabs_of |= (uint64_t)(S.dc_tab[cb] | (S.ac_tab[cb] << 2)) << (4u * b);
    }
    auto block_index = [&](uint32_t qq, uint32_t jj) { // decode-order block number -> index of its block word
        const uint32_t m = qq / bpm, mx = m % S.mcux, my = m / S.mcux, comp = S.blk_comp[jj];
        const JpegComponent &cc = c.H->comp[comp];
        return cc.block_base + (my * cc.v + S.blk_v[jj]) * cc.bw + mx * cc.h + S.blk_h[jj];
    };
    if (MODE == 2) gidx = q < S.total_blocks ? block_index(q, j) : 0u;
    while (p < p_end) {
        if (MODE == 2 && q >= S.total_blocks) break;
        if (cnt <= 32) { buf |= (uint64_t)next_word << (32 - cnt); cnt += 32; ++wi; next_word = word(wi); } // (the word after is requested at once: its latency hides behind the symbols in between)
        const uint32_t comp = (uint32_t)(comp_of >> (4u * j)) & 3u;
        const uint32_t tsel = (uint32_t)(tabs_of >> (4u * j)) & 15u;
        const uint32_t *tab = lut + (k == 0u ? (tsel & 3u) : (tsel >> 2)) * TW;
        const uint32_t e = tab[(uint32_t)(buf >> (64 - kJhLookBits))];
        uint32_t nb, run = 0, flags = 0; // flags: 1 = end of block / category 0, 2 = ZRL
        int32_t val = 0;
        if (e) {
            nb = e & 31u;
            if (e & (1u << 12)) flags = (e & (1u << 13)) ? 1u : 2u;
            else { run = (e >> 5) & 15u; val = (int32_t)(int16_t)(e >> 16); }
        } else {
            // code + magnitude longer than the lookahead: the code alone from the 9-bit table or the canonical search, then the magnitude
            const uint32_t f = (tab[T_LOOK + ((uint32_t)(buf >> 55) >> 1)] >> (16u * ((uint32_t)(buf >> 55) & 1u))) & 0xffffu;
            uint32_t len, sym;
            if (f) { len = f >> 8; sym = f & 255u; }
            else {
                len = 10u;
                const int32_t *maxcode = reinterpret_cast<const int32_t *>(tab + T_MAXCODE), *valoff = reinterpret_cast<const int32_t *>(tab + T_VALOFF);
                while (len <= 16u && (int32_t)(uint32_t)(buf >> (64u - len)) > maxcode[len]) ++len;
                if (len > 16u) { bad = true; len = 16u; sym = 0u; }
                else {
                    const int32_t idx = (int32_t)(uint32_t)(buf >> (64u - len)) + valoff[len];
                    if (idx < 0 || idx > 255) { bad = true; sym = 0u; }
                    else sym = (tab[T_VALS + ((uint32_t)idx >> 2)] >> (8u * ((uint32_t)idx & 3u))) & 255u;
                }
            }
            const uint32_t s = sym & 15u;
            run = sym >> 4;
            nb = len + s;
            if (s == 0u) flags = (run == 15u && k != 0u) ? 2u : 1u;
            else {
                const int32_t v = (int32_t)(uint32_t)((buf << len) >> (64u - s));
                val = v < (1 << (s - 1u)) ? v - (1 << s) + 1 : v;
            }
            if (k == 0u && (sym > 11u)) bad = true;            // a DC symbol is a bare category 0..11
            if (k != 0u && s == 0u && run != 0u && run != 15u) bad = true; // (run, 0) other than end of block / ZRL: not a baseline code
        }
        buf <<= nb; cnt -= (int)nb; p += nb;
        if (k == 0u) {
            if (flags == 2u || (flags == 0u && run != 0u)) bad = true;
            if (MODE >= 1 && comp < 3u) dcs[comp] += val;
            if (MODE == 2) {
                const int32_t dc = dc0[comp < 3u ? comp : 0u] + dcs[comp < 3u ? comp : 0u];
                if (dc < -32768 || dc > 32767) bad = true;
                coef[(size_t)gidx * 64u] = (int16_t)dc;
            }
            k = 1u;
        } else if (flags == 1u) k = 64u;
        else if (flags == 2u) k += 16u;
        else {
            k += run;
            if (k > 63u) { bad = true; k = 64u; }
            else {
                if (MODE == 2) coef[(size_t)gidx * 64u + k] = (int16_t)val;
                ++k;
            }
        }
        if (k >= 64u) { // block complete
            k = 0u;
            j = j + 1u == bpm ? 0u : j + 1u;
            ++nblk;
            if (MODE == 2) {
                if (bad) atomicOr(err, 2u); // (an invalid code word inside a real block: the file is broken, or the states were wrong)
                bad = false;
                ++q;
                if (q < S.total_blocks) gidx = block_index(q, j);
            }
        }
    }
    // (a block that straddles the subsequence's end: the next walk continues it with a clean flag, so what this part of it saw is
    // reported here -- the host decoder rejects the same file, and which of the two runs must not decide whether a corrupt file is served)
    if (MODE == 2 && bad) atomicOr(err, 2u);
    if (MODE >= 1 && cnt4) { cnt4[0] = nblk; cnt4[1] = dcs[0]; cnt4[2] = dcs[1]; cnt4[3] = dcs[2]; }
    return pack_state(p, j, k);
}

__global__ __launch_bounds__(256) void jh_init_kernel(const JhJob *jobs, uint32_t max_blocks)
{
    const JhJob jb = jobs[blockIdx.y];
    const JpegBlobHeader *H = reinterpret_cast<const JpegBlobHeader *>(jb.stage);
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {
        // header of the blob the IDCT kernel reads: the staged one, as "coefficients" (FJD1)
        for (uint32_t k = threadIdx.x; k < sizeof(JpegBlobHeader) / 4u; k += blockDim.x)
            reinterpret_cast<uint32_t *>(jb.blob)[k] = k == 0u ? 0x31444a46u : reinterpret_cast<const uint32_t *>(jb.stage)[k];
        if (threadIdx.x == 0) { *jb.err = 0u; jb.states[0] = 0ull; }
    }
    if (b >= H->nblocks) return;
    reinterpret_cast<uint32_t *>(jb.blob + H->blocks_off)[b] = ((b * 64u) << 7) | (63u << 1) | 1u;
    uint4 *z = reinterpret_cast<uint4 *>(jb.blob + H->coef_off + (size_t)b * 128u);
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = uint4{0u, 0u, 0u, 0u};
}

// Speculative decoding and re-synchronisation.  A workgroup holds the states of its 256 subsequences in LDS and iterates on them:
// in every round a subsequence whose START state (the end state of the one before it) differs from the one it last decoded from is
// decoded again, and its end state replaced if it changed; the rounds end when nothing changed (or after kJhInnerRounds).  A decoder
// that starts at a wrong bit finds the code-word boundaries again within a few symbols, the position inside the block at the next
// end-of-block code -- but its idea of WHICH block of the MCU it is in (i.e. which code tables apply) only falls into step through the
// garbage it decodes where luma and chroma tables differ, which can take several MCUs: hence rounds, not one pass.  The chain
// across workgroups moves once per launch (FIRST, then kJhSyncRounds more); the counting pass checks the result.
constexpr int kJhInnerRounds = 24;

template <bool FIRST>
__global__ __launch_bounds__(256) void jh_sync_kernel(const JhJob *jobs, const JhItem *items)
{
    __shared__ uint32_t lut[4 * TW];
    __shared__ uint32_t win[kWinWords];
    __shared__ uint64_t st[257];
    __shared__ int changed;
    const JhItem it = items[blockIdx.x];
    const JhJob jb = jobs[it.job];
    Ctx c = make_ctx(jb);
    stage_tables(jb, c, lut);
    stage_window(c, it.first_sub, win);
    const uint32_t t = threadIdx.x, sub = it.first_sub + t;
    const bool active = sub < jb.nsub;
    const uint32_t p_end = (sub + 1u) * kJhSubBits;
    uint64_t last_in = ~0ull; // the start state this subsequence was last decoded from
    int32_t cnt4[4] = {0, 0, 0, 0}; // ... and what that walk counted: blocks completed, DC difference sums per component
    bool walked = false;
    if (FIRST) {
        if (t == 0) st[0] = pack_state(it.first_sub * kJhSubBits, 0u, 0u); // (exact for the picture's first subsequence, a guess for every other workgroup)
        if (active) {
            last_in = pack_state(sub * kJhSubBits, 0u, 0u);
            st[t + 1u] = jh_walk<1>(c, lut, last_in, p_end, cnt4, 0u, nullptr, nullptr, nullptr);
            walked = true;
        }
    } else {
        if (t == 0) st[0] = jb.states[it.first_sub];
        if (active) { st[t + 1u] = jb.states[sub + 1u]; last_in = jb.used[sub]; }
    }
    for (int r = 0; r < kJhInnerRounds; ++r) {
        if (t == 0) changed = 0;
        __syncthreads();
        const uint64_t start = st[t];
        uint64_t end = 0;
        bool redo = false;
        if (active && start != last_in) {
            // (a start beyond this subsequence -- a walk that ran through it -- just passes on)
            if ((uint32_t)start >= p_end) { end = start; cnt4[0] = cnt4[1] = cnt4[2] = cnt4[3] = 0; }
            else end = jh_walk<1>(c, lut, start, p_end, cnt4, 0u, nullptr, nullptr, nullptr);
            walked = true;
            last_in = start;
            redo = end != st[t + 1u];
        }
        __syncthreads(); // every start of this round has been read
        if (redo) { st[t + 1u] = end; changed = 1; }
        __syncthreads();
        if (!changed) break;
    }
    if (active) {
        jb.states[sub + 1u] = st[t + 1u];
        jb.used[sub] = last_in;
        if (walked) {
#pragma unroll
            for (int k = 0; k < 4; ++k) jb.counts[sub * 4u + k] = cnt4[k];
        }
    }
}

// one workgroup per picture: checks that the chain of states is consistent -- every subsequence was last decoded from the end state
// its predecessor has now (else error bit 1: the states had not settled, the host decodes this file) -- and forms the exclusive
// prefix sums of counts[nsub][4]
__global__ __launch_bounds__(256) void jh_scan_kernel(const JhJob *jobs)
{
    __shared__ int32_t part[256][4];
    const JhJob jb = jobs[blockIdx.x];
    const uint32_t per = (jb.nsub + 255u) / 256u, lo = threadIdx.x * per, hi = min(lo + per, jb.nsub);
    int32_t s[4] = {0, 0, 0, 0};
    bool unsettled = false;
    for (uint32_t i = lo; i < hi; ++i) {
        unsettled |= jb.used[i] != jb.states[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += jb.counts[i * 4u + k];
    }
    if (unsettled) atomicOr(jb.err, 1u);
#pragma unroll
    for (int k = 0; k < 4; ++k) part[threadIdx.x][k] = s[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t run[4] = {0, 0, 0, 0};
        for (int t = 0; t < 256; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int32_t v = part[t][k]; part[t][k] = run[k]; run[k] += v; }
        // The segment must hold every block of the scan.  A truncated file (or one cut by a stray marker) whose last walk stops at the
        // segment's end before it meets the padding raises no invalid-code error, and its missing blocks would stay zero -- the host
        // decoder feeds zero bits past the end and carries the DC predictors on, i.e. decodes different pixels: error bit 4, the host
        // decodes this file.
        const JpegHuffStage *S = reinterpret_cast<const JpegHuffStage *>(jb.stage + sizeof(JpegBlobHeader));
        if (run[0] < (int32_t)S->total_blocks) atomicOr(jb.err, 4u);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = part[threadIdx.x][k];
    for (uint32_t i = lo; i < hi; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) { jb.prefix[i * 4u + k] = s[k]; s[k] += jb.counts[i * 4u + k]; }
}

__global__ __launch_bounds__(256) void jh_write_kernel(const JhJob *jobs, const JhItem *items)
{
    __shared__ uint32_t lut[4 * TW];
    __shared__ uint32_t win[kWinWords];
    const JhItem it = items[blockIdx.x];
    const JhJob jb = jobs[it.job];
    Ctx c = make_ctx(jb);
    stage_tables(jb, c, lut);
    stage_window(c, it.first_sub, win);
    const uint32_t sub = it.first_sub + threadIdx.x;
    if (sub >= jb.nsub) return;
    const uint32_t p_end = (sub + 1u) * kJhSubBits;
    const uint64_t start = jb.states[sub];
    if ((uint32_t)start >= p_end) return;
    const int32_t dc0[3] = {jb.prefix[sub * 4u + 1u], jb.prefix[sub * 4u + 2u], jb.prefix[sub * 4u + 3u]};
    int16_t *coef = reinterpret_cast<int16_t *>(jb.blob + c.H->coef_off);
    (void)jh_walk<2>(c, lut, start, p_end, nullptr, (uint32_t)jb.prefix[sub * 4u], dc0, coef, jb.err);
}

} // namespace

size_t jh_blob_bytes(const JpegBlobHeader &H) { return (size_t)H.coef_off + (size_t)H.nblocks * 128u + 64u; }
uint32_t jh_subsequences(const JpegHuffStage &S) { return (S.stream_bits + kJhSubBits - 1u) / kJhSubBits; }

hipError_t launch_jpeg_huff(const JhJob *d_jobs, const JhJob *h_jobs, uint32_t njobs, const JhItem *d_items, uint32_t nitems, uint32_t max_blocks, hipStream_t st)
{
    (void)h_jobs;
    if (!njobs || !nitems) return hipSuccess;
    jh_init_kernel<<<dim3((max_blocks + 255u) / 256u, njobs), 256, 0, st>>>(d_jobs, max_blocks);
    jh_sync_kernel<true><<<nitems, 256, 0, st>>>(d_jobs, d_items);
    for (uint32_t r = 0; r < kJhSyncRounds; ++r) jh_sync_kernel<false><<<nitems, 256, 0, st>>>(d_jobs, d_items); // (each moves the chain across one more workgroup boundary)
    jh_scan_kernel<<<njobs, 256, 0, st>>>(d_jobs);
    jh_write_kernel<<<nitems, 256, 0, st>>>(d_jobs, d_items);
    return hipGetLastError();
}

} // namespace fl
