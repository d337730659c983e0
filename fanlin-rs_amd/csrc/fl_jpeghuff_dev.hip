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
//               exact; self-synchronisation does the rest.  Measured on a 4:2:0 photograph (FL_JH_TRACE, tools/experiments/jh_trace.py):
//               code words and the position inside the block fall into step within a block or two, but WHICH block of the MCU
//               (luma or chroma tables) only by chance -- about one time in six per MCU -- so a workgroup of 256 subsequences
//               iterates 5..9 times (255, ~100, ~35, ~17, ... subsequences decoded again per round) and the kernel's time is
//               rounds x one walk (53 us), whatever the batch
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
constexpr uint32_t LB = kJhLookBits;
constexpr uint32_t T_MAXCODE = (1u << LB) / 2u, T_VALOFF = T_MAXCODE + 18, T_VALS = T_VALOFF + 17; // word offsets inside a table
static_assert(T_VALS + 64u <= TW, "device code table layout (fl_jpegdec.h kJhTableWords)");

struct Ctx {
    const JpegBlobHeader *H;
    const JpegHuffStage *S;
    const uint32_t *words;  // the unstuffed segment as big-endian words
    uint32_t nwords;
    const uint32_t *lwords; // the workgroup's window of it in LDS, byte-swapped: words [lbase, lbase + kWinWords)
    uint32_t lbase;
    uint32_t bpm, total_blocks, mcux;
    uint32_t rst_blocks;    // blocks per restart interval (0: the file has none)
    uint32_t n_rst;
    const uint32_t *rst;    // byte offsets (in the segment) at which the intervals after the first start, ascending
};
// The speculative kernel's workgroups also decode the kJhWarm subsequences IN FRONT of their own kJhSubsPerItem (results discarded): by
// the time the chain of states reaches the workgroup's own first subsequence it has fallen into step (a chain longer than sixteen
// subsequences: (5/6)^51, one in 10^4 workgroups -- the later launches and the scan are still there for it), so the launches that used
// to carry the chain across workgroup fronts find nothing to do and leave at once (148 -> 18 us each, profiles/r05_jpeg_decode_kernels.txt).
// 240 + 16 = the four waves of a workgroup: a fifth wave for the warm-up shared a SIMD and cost the speculative kernel 12 %.
constexpr uint32_t kJhOwn = kJhSubsPerItem;
constexpr uint32_t kJhWarm = 256u - kJhOwn;
constexpr uint32_t kJhMaxSubs = 256u;
constexpr uint32_t kWinWords = kJhMaxSubs * kJhSubBits / 32u + 8u; // the subsequences + the words a walk may read past its end

// LDS of the walking kernels (dynamic: 4 tables + window + per-block records + states = 69 KB, beyond the static 64 KB)
constexpr uint32_t kLdsLut = 0, kLdsWin = kLdsLut + 4u * TW, kLdsBinfo = kLdsWin + kWinWords, kLdsStates = kLdsBinfo + 12u * 4u, kLdsRst = kLdsStates + 2u * (kJhMaxSubs + 2u), kRstWords = kWinWords / 8u + 2u, kLdsRstCnt = kLdsRst + kRstWords, kLdsWords = kLdsRstCnt + kRstWords + 4u;
static_assert(kLdsStates % 2u == 0u && kLdsBinfo % 4u == 0u, "LDS alignment of the 64-bit states / 16-byte block records");

__device__ __forceinline__ Ctx make_ctx(const JhJob &jb)
{
    Ctx c;
    c.H = reinterpret_cast<const JpegBlobHeader *>(jb.stage);
    c.S = reinterpret_cast<const JpegHuffStage *>(jb.stage + sizeof(JpegBlobHeader));
    c.words = reinterpret_cast<const uint32_t *>(jb.stage + c.S->stream_off);
    c.nwords = (c.S->stream_bits / 8u + 16u) / 4u; // (the stage is padded with 16 bytes of ones)
    c.lwords = nullptr; c.lbase = 0u;
    c.bpm = c.S->bpm; c.total_blocks = c.S->total_blocks; c.mcux = c.S->mcux;
    c.rst_blocks = c.S->rst_mcus * c.S->bpm; c.n_rst = c.S->n_rst;
    c.rst = reinterpret_cast<const uint32_t *>(jb.stage + c.S->rst_off);
    return c;
}

// The 256 subsequences of a workgroup are one contiguous piece of the segment: it is copied to LDS once (coalesced, already in the
// byte order the bit buffer wants), and the walks read their words there -- a walk consumes a word every five symbols or so, and
// fetched one by one from the L2 those loads were most of its time.
__device__ __forceinline__ void stage_window(Ctx &c, uint32_t first_sub, uint32_t *win, uint32_t nsubs = kJhOwn)
{
    const uint32_t base = first_sub * (kJhSubBits / 32u), nwords = nsubs * (kJhSubBits / 32u) + 8u;
    for (uint32_t k = threadIdx.x; k < nwords; k += blockDim.x) {
        const uint32_t i = base + k;
        win[k] = __builtin_bswap32(c.words[i < c.nwords ? i : c.nwords - 1u]);
    }
    c.lwords = win; c.lbase = base;
    __syncthreads();
}

// the picture's four code tables, and per block of the MCU where its block words are: {first, step per MCU column, step per MCU row}
__device__ __forceinline__ void stage_tables(const JhJob &jb, const Ctx &c, uint32_t *lut, uint32_t *binfo)
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(jb.stage + c.S->tables_off);
    for (uint32_t k = threadIdx.x; k < 4u * TW; k += blockDim.x) lut[k] = src[k];
    if (threadIdx.x < c.bpm) {
        const uint32_t jj = threadIdx.x, cb = c.S->blk_comp[jj];
        const JpegComponent &cc = c.H->comp[cb];
        // what a walk needs of block jj of the MCU: its AC and DC table (byte offsets in LDS), its component as the shift of its field in the
        // DC sums; and where its block words are: {first, step per MCU column, step per MCU row}
        binfo[4u * jj] = (uint32_t)c.S->ac_tab[cb] * (4u * TW) | ((uint32_t)c.S->dc_tab[cb] * (4u * TW)) << 16;
        binfo[4u * jj + 1u] = 21u * cb;
        binfo[4u * jj + 2u] = cc.block_base + c.S->blk_v[jj] * cc.bw + c.S->blk_h[jj];
        binfo[4u * jj + 3u] = cc.h | (cc.v * cc.bw) << 16; // (a component's block rows are at most 8192 x 2 blocks long)
    }
    __syncthreads();
}

// Restart intervals: one bit per byte of the workgroup's window, set where an interval starts.  (A walk asks "does an interval start at the
// next byte?" at every block start; the list itself is per picture and would need a search per walk.)
__device__ __forceinline__ void stage_restarts(const Ctx &c, uint32_t first_sub, uint32_t *rbits, uint32_t nsubs = kJhOwn)
{
    // rbits[kRstWords]: the bitmap; behind it rcnt[kRstWords]: interval starts in the window in front of each word; rcnt[kRstWords]: ... in front of the window
    uint32_t *rcnt = rbits + kRstWords;
    for (uint32_t k = threadIdx.x; k < kRstWords; k += blockDim.x) rbits[k] = 0u;
    if (threadIdx.x == 0) rcnt[kRstWords] = 0u;
    __syncthreads();
    const uint32_t base = first_sub * (kJhSubBits / 8u), nbytes = nsubs * (kJhSubBits / 8u) + 32u;
    uint32_t before = 0;
    for (uint32_t i = threadIdx.x; i < c.n_rst; i += blockDim.x) {
        const uint32_t b = c.rst[i], rel = b - base;
        before += b < base ? 1u : 0u;
        if (rel < nbytes) atomicOr(&rbits[rel >> 5], 1u << (rel & 31u));
    }
    if (before) atomicAdd(&rcnt[kRstWords], before);
    __syncthreads();
    // exclusive prefix of the words' bit counts (the interval an interval start opens has a number: what the block counter must show there)
    constexpr uint32_t per = (kRstWords + 255u) / 256u;
    uint32_t local = 0;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t w = threadIdx.x * per + k; if (w < kRstWords) local += __popc(rbits[w]); }
    __shared__ uint32_t scan[256];
    scan[threadIdx.x] = local;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? scan[threadIdx.x - d] : 0u;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = scan[threadIdx.x] - local;
    for (uint32_t k = 0; k < per; ++k) { const uint32_t w = threadIdx.x * per + k; if (w < kRstWords) { rcnt[w] = run; run += __popc(rbits[w]); } }
    __syncthreads();
}

__device__ __forceinline__ uint64_t pack_state(uint32_t p, uint32_t j, uint32_t k) { return (uint64_t)p | ((uint64_t)j << 32) | ((uint64_t)k << 40); }

// One walk over the code words from state (p, j, k) until the bit position reaches p_end.
// MODE 1: states, blocks completed and DC difference sums.  MODE 2: + the coefficients are stored (q0 = number of the block the walk
// starts in, dc0 = the components' DC predictors there) and invalid code words of real blocks are reported.
//
// The 64 lanes of a wave are in 64 different places of their blocks, so every branch of the loop body is taken by SOME lane in almost
// every step: a step costs the sum of all its paths, and with one wave per SIMD walking a dependent chain the walk's time is its
// instruction count (measured: ~7 cycles per instruction, FL_JH_TRACE).  Hence ONE straight-line symbol path for every lane:
//   * a kJhLookBits lookahead gives (code length, symbol) for every code the files in practice use; the magnitude bits are taken
//     arithmetically (T.81 F.2.2.1 EXTEND without a branch); only codes longer than the lookahead (Annex K: the 15- and 16-bit
//     ones) leave the path, for the canonical search of F.2.2.3;
//   * the bit window is 32 bits cut from two register words (v_alignbit), the word after them requested in every step;
//   * the block's tables come from a record per block of the MCU (LDS), rotated cur <- next <- after-next when a block ends -- no
//     64-bit shifts per symbol, no division in the write pass (block numbers advance by increments);
//   * the NEXT symbol's table entry is requested -- from the block's AC table and the next block's DC table at once -- before this
//     symbol's bookkeeping, which then runs while the reads are under way;
//   * an end-of-block code that follows a symbol inside the lookahead is consumed with it.
// Round 4's walk resolved code + magnitude in one 10-bit lookup and sent everything longer -- one symbol in eight at quality 85 --
// through a second table and a search; with 64 lanes that path ran in every step: ~350 instructions per step, this one ~90
// (first kernel 911 -> 650 us, write pass 261 -> 155 us per batch of 17 files; profiles/r05_jpeg_decode_kernels.txt).
// RST: the file has restart intervals (rbits = the window's bitmap of interval starts).  At a block start with an interval starting at the
// next byte boundary, what is left of the byte is padding (F.1.2.3): the walk steps to the boundary, the DC predictors start again at zero
// (F.2.1.3.1) and the block counter must stand at a multiple of the interval -- for a walk that is out of step this is also where it
// falls into step for good: position, block of the MCU and coefficient index are all known there.
template <int MODE, bool RST>
__device__ __forceinline__ uint64_t jh_walk(const Ctx &c, const uint32_t *lut, const uint32_t *binfo, const uint32_t *rbits, uint64_t state, uint32_t p_end, int32_t *cnt4,
                                            uint32_t q0, const int32_t *dc0, int16_t *coef, uint32_t *err)
{
    static_assert(MODE == 1 || MODE == 2, "walk mode");
    const char *lds0 = reinterpret_cast<const char *>(lut); // (the tables start the workgroup's LDS block; block records hold BYTE offsets from here)
    auto look = [&](uint32_t byte_off) { return (uint32_t)*reinterpret_cast<const uint16_t *>(lds0 + byte_off); };
    auto word = [&](uint32_t i) { return c.lwords[i < kWinWords ? i : kWinWords - 1u]; };
    // per block of the MCU: {AC table | DC table << 16 (byte offsets in LDS), 21 * component, first block word, step per MCU column | per MCU row << 16}
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto info = [&](uint32_t jj) { return *reinterpret_cast<const u32x4 *>(binfo + 4u * jj); };
    auto after = [&](uint32_t jj) { return jj + 1u >= c.bpm ? jj + 1u - c.bpm : jj + 1u; };
    uint32_t j = (uint32_t)(state >> 32) & 15u, k = (uint32_t)(state >> 40) & 127u;
    if (j >= c.bpm) j = 0u; // (a state is only ever one a walk produced; this keeps a corrupt one inside the tables)
    // bit position inside the LDS window (a walk starts in or behind its own subsequence and ends within a word of its end: inside by construction)
    uint32_t pr = (uint32_t)state - 32u * c.lbase;
    const uint32_t pr_end = p_end - 32u * c.lbase;
    // the 32 bits at pr: words W and W + 1 of the window in registers, W + 2 requested in every step (it has arrived when a step crosses into W + 1)
    uint32_t W = pr >> 5;
    uint32_t hi = word(W), lo = word(W + 1u), nx = word(W + 2u);
    auto bits_at = [&](uint32_t pos) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (32u - (pos & 31u))); }; // (a shift by 1 .. 32: one instruction, no special case for a position on a word boundary)
    uint32_t win = bits_at(pr);
    // block records: of the block the walk is in, of the next one, and (requested in every step) of the one after -- a block can be one step long
    uint32_t j2 = after(after(j)); // the block the record in `nn` belongs to: two on from the walk's (one wrap-around per step instead of three)
    u32x4 cur = info(j), nxt = info(after(j)), nn = info(j2);
    int32_t nblk = 0;
    int64_t dacc = 0;        // MODE 1: the three DC difference sums as 21-bit signed fields of one 64-bit sum (|sum| < 2^20: at most 79 blocks of |difference| <= 2047 fit 1024 bits)
    int32_t d0 = 0, d1 = 0, d2 = 0; // MODE 2: the components' DC predictors
    uint32_t q = q0, gidx = 0, mx = 0, my = 0;
    bool bad = false;
    auto place = [&](const u32x4 &rec, uint32_t x, uint32_t y) { return rec.z + y * (rec.w >> 16) + x * (rec.w & 0xffffu); }; // block word of the record's block in MCU (x, y)
    if (MODE == 2) {
        d0 = dc0[0]; d1 = dc0[1]; d2 = dc0[2];
        const uint32_t m = q / c.bpm;
        my = m / c.mcux; mx = m - my * c.mcux;
        gidx = place(cur, mx, my);
    }
    // The table entry of the NEXT symbol is requested as soon as the window has moved past this one -- from the block's AC table and
    // from the next block's DC table at once, since which of the two it is (did this symbol end the block?) comes out of the
    // bookkeeping that runs while the reads are under way.  The recurrence of a step is then bits -> entry -> lengths -> bits.
    uint32_t e = look((k == 0u ? cur.x >> 16 : cur.x & 0xffffu) + 2u * (win >> (32u - LB)));
    // the bitmap words that hold the next byte boundary and the four bytes behind it (requested with the table entries); the boundary stepped to last
    auto rpair_at = [&](uint32_t pos) { const uint32_t w = ((pos + 7u) >> 3) >> 5; return (uint64_t)rbits[w] | (uint64_t)rbits[w + 1u] << 32; };
    uint64_t rpair = RST ? rpair_at(pr) : 0ull;
    uint32_t snapped = 0xffffffffu;
    bool restarted = false;
    while (pr < pr_end && (MODE != 2 || q < c.total_blocks)) {
#ifdef FL_JH_TRACE
        if (MODE == 1 && cnt4) ++cnt4[4];
#endif
        // interval starts at the next byte boundary (bit 0) and at the four bytes behind it (a step moves 31 bits at most): zero almost always
        const uint32_t nbyte = (pr + 7u) >> 3, near = RST ? (uint32_t)(rpair >> (nbyte & 31u)) & 0x1fu : 0u;
        if (RST && (near & 1u)) {
            // An interval starts at the next byte boundary, the walk stands at a block start, and what is left of the byte is all ones: padding
            // (F.1.2.3).  The ones matter: the interval's LAST block may well start inside its last byte ("DC difference 0, end of block" is four
            // bits) -- but no block starts with ones only, because no Huffman code is all ones (K.2: the all-ones code word is never assigned).
            const uint32_t rem = (nbyte << 3) - pr;
            if (k == 0u && nbyte != snapped && (rem == 0u || (~win >> (32u - rem)) == 0u)) {
                snapped = nbyte;
                if (MODE == 2) { // the interval that starts here has a number, and the block counter must stand at exactly that many intervals' blocks
                    const uint32_t *rcnt = rbits + kRstWords;
                    const uint32_t index = rcnt[kRstWords] + rcnt[nbyte >> 5] + (uint32_t)__popc((uint32_t)rpair & ((1u << (nbyte & 31u)) - 1u)); // interval starts in front of this one
                    if (j != 0u || q != (index + 1u) * c.rst_blocks) bad = true; // (the host decoder wants its marker exactly behind the interval's last MCU)
                }
                pr = nbyte << 3; j = 0u;
                j2 = after(after(0u));
                cur = info(0u); nxt = info(after(0u)); nn = info(j2);
                W = pr >> 5; hi = word(W); lo = word(W + 1u); nx = word(W + 2u);
                win = bits_at(pr);
                e = look((cur.x >> 16) + 2u * (win >> (32u - LB)));
                dacc = 0; restarted = true; d0 = d1 = d2 = 0;
                if (MODE == 2) gidx = place(cur, mx, my);
                continue; // (the boundary may be this walk's end)
            }
            // (a block that is still open where an interval starts: the host decoder finds no marker behind the interval's last MCU and rejects the file)
            if (MODE == 2 && k != 0u && rem == 0u) bad = true;
        }
        const uint32_t pr_before = pr;
        const bool isdc = k == 0u;
        // entry: bits 0-4 code length (0: longer than the lookahead), 5-7 length of an END-OF-BLOCK code that follows the symbol's
        // magnitude bits inside the lookahead (0: none there) -- it is consumed with the symbol: one step for the "DC difference, end of
        // block" blocks of flat regions, one step less for every block whose last coefficient is a short code --, 8-15 the symbol
        uint32_t len = e & 31u, sym = e >> 8;
        if (len == 0u) {
            // a code longer than the lookahead: canonical search (F.2.2.3) from the next length on
            const uint32_t *tab = reinterpret_cast<const uint32_t *>(lds0 + (isdc ? cur.x >> 16 : cur.x & 0xffffu));
            const int32_t *maxcode = reinterpret_cast<const int32_t *>(tab + T_MAXCODE), *valoff = reinterpret_cast<const int32_t *>(tab + T_VALOFF);
            len = LB + 1u;
            while (len <= 16u && (int32_t)(win >> (32u - len)) > maxcode[len]) ++len;
            sym = 0u;
            if (len > 16u) { bad = true; len = 16u; }
            else {
                const int32_t idx = (int32_t)(win >> (32u - len)) + valoff[len];
                if (idx < 0 || idx > 255) bad = true;
                else sym = (tab[T_VALS + ((uint32_t)idx >> 2)] >> (8u * ((uint32_t)idx & 3u))) & 255u;
            }
        }
        const uint32_t s = sym & 15u, run = sym >> 4, kr = k + run;
        // (coefficient 63 ends its block without an end-of-block code: what follows it is the next block's DC code)
        const uint32_t el = kr >= 63u ? 0u : (e >> 5) & 7u;
        const uint32_t t = win << len; // the bits behind the code
        pr += len + s + el;            // <= 16 + 15 bits (el != 0: <= the lookahead)
        const uint32_t Wn = pr >> 5;
        hi = Wn != W ? lo : hi; lo = Wn != W ? nx : lo; W = Wn;
        nx = word(Wn + 2u);
        win = bits_at(pr);
        const uint32_t nidx = 2u * (win >> (32u - LB));
        const uint32_t e_ac = look((cur.x & 0xffffu) + nidx), e_dc = look((nxt.x >> 16) + nidx);
        if (RST) rpair = rpair_at(pr);
        if (RST && MODE == 2 && near) {
            // A code word must not reach across an interval start (pr_before < 8 b < pr for a start byte b): the walk did not step to it -- the
            // bits in front of it were not padding, or the interval holds more than its MCUs -- and what follows would be decoded with the DC
            // predictors of the interval before.  The byte boundaries inside a step's 31 bits are among the five `near` covers.
            const uint32_t b0 = (pr_before >> 3) + 1u, b1 = (pr - 1u) >> 3;
            if (b1 >= b0 && ((near >> (b0 - nbyte)) & ((1u << (b1 - b0 + 1u)) - 1u))) bad = true;
        }
        // RECEIVE + EXTEND: the s bits behind the code; a leading 0 bit means negative, value - 2^s + 1 (s = 0: no bits, 0)
        const int32_t val = (int32_t)((t >> 1) >> (31u - s)) + (((int32_t)t >> 31) ? 0 : (int32_t)((0xffffffffu << s) + 1u));
        if (MODE == 1) dacc += (int64_t)(isdc ? val : 0) << cur.y;
        if (MODE == 2) {
            const uint32_t comp = cur.y == 0u ? 0u : cur.y == 21u ? 1u : 2u;
            const int32_t dc = (comp == 0u ? d0 : comp == 1u ? d1 : d2) + val;
            if (isdc) { d0 = comp == 0u ? dc : d0; d1 = comp == 1u ? dc : d1; d2 = comp == 2u ? dc : d2; }
            // a DC symbol is a bare category 0..11 and the predictor stays 16-bit; (run, 0) other than end of block / ZRL is not a baseline code; no coefficient 64
            bad |= isdc ? (sym > 11u || dc < -32768 || dc > 32767) : (s == 0u ? (run != 0u && run != 15u) : kr > 63u);
            // (a GLOBAL store with a 32-bit index: a flat one also counts as an LDS operation, and the wait for the next table entry at the
            // top of the loop then waited for the store's trip to memory as well -- the write pass ran at a third of the counting walks' speed)
            if (isdc || (s != 0u && kr <= 63u)) ((__attribute__((address_space(1))) int16_t *)coef)[gidx * 64u + (isdc ? 0u : kr)] = (int16_t)(isdc ? dc : val);
        }
        const uint32_t k_ac = s == 0u ? (run == 15u ? k + 16u : 64u) : (el ? 64u : kr + 1u);
        k = isdc ? (el ? 64u : 1u) : k_ac;
        const bool done = k >= 64u; // block complete
        k = done ? 0u : k;
        nblk += done ? 1 : 0;
        {
            const uint32_t j3 = after(j2);
            j = done ? (j + 1u == c.bpm ? 0u : j + 1u) : j;
            j2 = done ? j3 : j2;
        }
        if (MODE == 2) { // (straight-line like the rest: the next block's place from the record already in registers, no wait inside a branch)
            const bool wrap = done && j == 0u;                       // the block that starts now is the first of the next MCU
            const bool row = wrap && mx + 1u == c.mcux;
            mx = row ? 0u : mx + (wrap ? 1u : 0u);
            my += row ? 1u : 0u;
            gidx = done ? place(nxt, mx, my) : gidx;
            q += done ? 1u : 0u;
        }
        cur.x = done ? nxt.x : cur.x; cur.y = done ? nxt.y : cur.y;
        nxt.x = done ? nn.x : nxt.x; nxt.y = done ? nn.y : nxt.y;
        if (MODE == 2) { nxt.z = done ? nn.z : nxt.z; nxt.w = done ? nn.w : nxt.w; }
        nn = info(j2);
        e = k == 0u ? e_dc : e_ac; // (e_dc was read from what was then the next block's table: this block's, if the block has just changed)
    }
    // Invalid code words, a DC term out of range, a coefficient past 63 -- in any block of the scan this walk decoded (it stops at the scan's
    // last block: the padding behind it is never looked at): the file is broken, or the states were wrong; the host decodes it, and rejects it
    // if it is the file.  (Reported once per walk: a block that straddles the subsequence's end is continued by the next walk with its own flag.)
    if (MODE == 2 && bad) atomicOr(err, 2u);
    if (MODE == 1 && cnt4) {
        const int32_t c0 = (int32_t)((dacc << 43) >> 43);
        const int64_t a1 = (dacc - c0) >> 21;
        const int32_t c1 = (int32_t)((a1 << 43) >> 43);
        // (bit 31 of the block count: the walk passed an interval start, and the sums are those of the differences behind the last one)
        cnt4[0] = nblk | (restarted ? (int32_t)0x80000000 : 0); cnt4[1] = c0; cnt4[2] = c1; cnt4[3] = (int32_t)((a1 - c1) >> 21);
    }
    return pack_state(pr + 32u * c.lbase, j, k);
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
    const uint32_t nblocks = H->nblocks, first = blockIdx.x * blockDim.x;
    if (first >= nblocks) return;
    if (b < nblocks) reinterpret_cast<uint32_t *>(jb.blob + H->blocks_off)[b] = ((b * 64u) << 7) | (63u << 1) | 1u;
    // the coefficients of the workgroup's 256 blocks are one stretch of 32 KB: consecutive lanes zero consecutive 16 bytes (a lane zeroing
    // its own block's 128 bytes touched 64 different lines per store instruction: 32 us per batch of 13 files)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __attribute__((address_space(1))) u32x4 *z = (__attribute__((address_space(1))) u32x4 *)(uintptr_t)(jb.blob + H->coef_off + (size_t)first * 128u);
    const uint32_t n16 = min(256u, nblocks - first) * 8u; // 16-byte pieces
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k)
        if (k * 256u + threadIdx.x < n16) z[k * 256u + threadIdx.x] = u32x4{0u, 0u, 0u, 0u};
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
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *lut = lds + kLdsLut, *win = lds + kLdsWin, *binfo = lds + kLdsBinfo, *rbits = lds + kLdsRst;
    uint64_t *st = reinterpret_cast<uint64_t *>(lds + kLdsStates);
    __shared__ int changed;
    const JhItem it = items[blockIdx.x];
    const JhJob jb = jobs[it.job];
    // FIRST: threads [0, warm) decode the subsequences in front of the workgroup's own (none in front of a picture's first workgroup)
    const uint32_t warm = FIRST ? min(kJhWarm, it.first_sub) : 0u, base_sub = it.first_sub - warm;
    const uint32_t t = threadIdx.x, sub = base_sub + t;
    const bool active = sub < jb.nsub && t < kJhOwn + warm, own = active && t >= warm;
    // a later launch has work in a workgroup only where a subsequence's start state is no longer the one it was decoded from (the
    // chain has moved across the workgroup's front, or the rounds before ran out): everyone else leaves before the 77 KB of set-up
    if (!FIRST && !__syncthreads_or(active && jb.used[sub] != jb.states[sub])) return;
    Ctx c = make_ctx(jb);
    stage_tables(jb, c, lut, binfo);
    stage_window(c, base_sub, win, kJhOwn + warm);
    // (uniform in a workgroup: it serves one picture.  Files without restart intervals run the loop without the interval-start look-ups.)
    const bool rst = c.n_rst != 0u;
    if (rst) stage_restarts(c, base_sub, rbits, kJhOwn + warm);
    auto walk = [&](uint64_t from, uint32_t to, int32_t *cnt) {
        return rst ? jh_walk<1, true>(c, lut, binfo, rbits, from, to, cnt, 0u, nullptr, nullptr, nullptr) : jh_walk<1, false>(c, lut, binfo, rbits, from, to, cnt, 0u, nullptr, nullptr, nullptr);
    };
    const uint32_t p_end = (sub + 1u) * kJhSubBits;
    uint64_t last_in = ~0ull; // the start state this subsequence was last decoded from
#ifdef FL_JH_TRACE
    int32_t cnt4[5] = {0, 0, 0, 0, 0};
#else
    int32_t cnt4[4] = {0, 0, 0, 0}; // ... and what that walk counted: blocks completed, DC difference sums per component
#endif
    bool walked = false;
#ifdef FL_JH_TRACE
    const uint64_t tr0 = wall_clock64();
    const uint64_t cy0 = clock64();
    uint64_t tr1 = tr0;
    int tr_rounds = 0, tr_active[32] = {};
#endif
    if (FIRST) {
        if (t == 0) st[0] = pack_state(base_sub * kJhSubBits, 0u, 0u); // (exact for the picture's first subsequence, a guess for every other workgroup)
        if (active) {
            last_in = pack_state(sub * kJhSubBits, 0u, 0u);
            st[t + 1u] = walk(last_in, p_end, cnt4);
            walked = true;
        }
    } else {
        if (t == 0) st[0] = jb.states[base_sub];
        if (active) { st[t + 1u] = jb.states[sub + 1u]; last_in = jb.used[sub]; }
    }
#ifdef FL_JH_TRACE
    tr1 = wall_clock64();
    const uint64_t cy1 = clock64();
    __shared__ int tr_max, tr_sum;
    if (t == 0) { tr_max = 0; tr_sum = 0; }
    __syncthreads();
    atomicMax(&tr_max, cnt4[4]); atomicAdd(&tr_sum, cnt4[4]);
    __syncthreads();
    if (t == 0 && blockIdx.x < 12) printf("jh_sync<%d> wg %u: first walk %.1f us = %llu shader cycles, steps max %d mean %d\n", (int)FIRST, blockIdx.x, (double)(tr1 - tr0) / 100.0, (unsigned long long)(cy1 - cy0), tr_max, tr_sum / 256);
#endif
    for (int r = 0; r < kJhInnerRounds; ++r) {
#ifdef FL_JH_TRACE
        tr_rounds = r + 1;
        tr_active[r] = __syncthreads_count(active && st[t] != last_in);
#endif
        if (t == 0) changed = 0;
        __syncthreads();
        const uint64_t start = st[t];
        uint64_t end = 0;
        bool redo = false;
        if (active && start != last_in) {
            // (a start beyond this subsequence -- a walk that ran through it -- just passes on)
            if ((uint32_t)start >= p_end) { end = start; cnt4[0] = cnt4[1] = cnt4[2] = cnt4[3] = 0; }
            else end = walk(start, p_end, cnt4);
            walked = true;
            last_in = start;
            redo = end != st[t + 1u];
        }
        __syncthreads(); // every start of this round has been read
        if (redo) { st[t + 1u] = end; changed = 1; }
        __syncthreads();
        if (!changed) break;
    }
#ifdef FL_JH_TRACE
    if (t == 0 && blockIdx.x < 12) {
        const uint64_t tr2 = wall_clock64();
        printf("jh_sync<%d> wg %u: set-up+first walk %.1f us, %d rounds %.1f us; active per round:", (int)FIRST, blockIdx.x, (double)(tr1 - tr0) / 100.0, tr_rounds, (double)(tr2 - tr1) / 100.0);
        for (int r = 0; r < tr_rounds; ++r) printf(" %d", tr_active[r]);
        printf("\n");
    }
#endif
    if (own) {
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
    // per thread: blocks, the three DC sums, and whether a restart lies in its stretch (then the sums are those behind the last one:
    // bit 31 of a subsequence's block count says its walk passed an interval start, fl_jpeghuff_dev.hip jh_walk).  Stretches combine as
    // (a, b) -> blocks a + b, sums b's own if b holds a restart, else a + b: associative, so the 256 stretches are scanned in eight doubling
    // steps (until round 5 thread 0 walked over them: 256 dependent steps, half of the kernel's 32 us).
    __shared__ int32_t part[256][5];
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    typedef i32x4 __attribute__((aligned(4))) i32x4_a4;
    const JhJob jb = jobs[blockIdx.x];
    const uint32_t per = (jb.nsub + 255u) / 256u, lo = min(threadIdx.x * per, jb.nsub), hi = min(lo + per, jb.nsub);
    const i32x4_a4 *counts = reinterpret_cast<const i32x4_a4 *>(jb.counts);
    i32x4_a4 *prefix = reinterpret_cast<i32x4_a4 *>(jb.prefix);
    auto add = [&](int32_t (&s)[4], const i32x4 &cnt, int32_t *any) {
        const bool restarted = cnt.x < 0;
        s[0] += cnt.x & 0x7fffffff;
        s[1] = restarted ? cnt.y : s[1] + cnt.y; s[2] = restarted ? cnt.z : s[2] + cnt.z; s[3] = restarted ? cnt.w : s[3] + cnt.w;
        if (any && restarted) *any = 1;
    };
    int32_t s[4] = {0, 0, 0, 0}, any = 0;
    bool unsettled = false;
    for (uint32_t i = lo; i < hi; i += 4u) { // (four subsequences' loads under way at once: the stretch is ~10 long and every load a trip to L2)
        i32x4 cn[4]; uint64_t us[4], st[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t ii = min(i + u, hi - 1u);
            cn[u] = counts[ii]; us[u] = jb.used[ii]; st[u] = jb.states[ii];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            if (i + u < hi) { unsettled |= us[u] != st[u]; add(s, cn[u], &any); }
    }
    if (unsettled) atomicOr(jb.err, 1u);
    int32_t mine[5] = {s[0], s[1], s[2], s[3], any};
#pragma unroll
    for (int k = 0; k < 5; ++k) part[threadIdx.x][k] = mine[k];
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) { // inclusive scan
        int32_t l[5] = {0, 0, 0, 0, 0};
        const bool has = threadIdx.x >= d;
        if (has) {
#pragma unroll
            for (int k = 0; k < 5; ++k) l[k] = part[threadIdx.x - d][k];
        }
        __syncthreads();
        if (has) {
            mine[0] += l[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) mine[k] = mine[4] ? mine[k] : l[k] + mine[k];
            mine[4] |= l[4];
#pragma unroll
            for (int k = 0; k < 5; ++k) part[threadIdx.x][k] = mine[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 255u) {
        // The segment must hold every block of the scan.  A truncated file (or one cut by a stray marker) whose last walk stops at the
        // segment's end before it meets the padding raises no invalid-code error, and its missing blocks would stay zero -- the host
        // decoder feeds zero bits past the end and carries the DC predictors on, i.e. decodes different pixels: error bit 4, the host
        // decodes this file.
        const JpegHuffStage *S = reinterpret_cast<const JpegHuffStage *>(jb.stage + sizeof(JpegBlobHeader));
        if (mine[0] < (int32_t)S->total_blocks) atomicOr(jb.err, 4u);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = threadIdx.x ? part[threadIdx.x - 1u][k] : 0; // exclusive: everything in front of this stretch
    for (uint32_t i = lo; i < hi; i += 4u) {
        i32x4 cn[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) cn[u] = counts[min(i + u, hi - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u)
            if (i + u < hi) {
                prefix[i + u] = i32x4{s[0], s[1], s[2], s[3]};
                add(s, cn[u], nullptr);
            }
    }
}

__global__ __launch_bounds__(256) void jh_write_kernel(const JhJob *jobs, const JhItem *items)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *lut = lds + kLdsLut, *win = lds + kLdsWin, *binfo = lds + kLdsBinfo, *rbits = lds + kLdsRst;
    const JhItem it = items[blockIdx.x];
    const JhJob jb = jobs[it.job];
    Ctx c = make_ctx(jb);
    stage_tables(jb, c, lut, binfo);
    stage_window(c, it.first_sub, win);
    const bool rst = c.n_rst != 0u;
    if (rst) stage_restarts(c, it.first_sub, rbits);
    const uint32_t sub = it.first_sub + threadIdx.x;
    if (threadIdx.x >= kJhOwn || sub >= jb.nsub) return;
    const uint32_t p_end = (sub + 1u) * kJhSubBits;
    const uint64_t start = jb.states[sub];
    if ((uint32_t)start >= p_end) return;
    const int32_t dc0[3] = {jb.prefix[sub * 4u + 1u], jb.prefix[sub * 4u + 2u], jb.prefix[sub * 4u + 3u]};
    int16_t *coef = reinterpret_cast<int16_t *>(jb.blob + c.H->coef_off);
    if (rst) (void)jh_walk<2, true>(c, lut, binfo, rbits, start, p_end, nullptr, (uint32_t)jb.prefix[sub * 4u], dc0, coef, jb.err);
    else (void)jh_walk<2, false>(c, lut, binfo, rbits, start, p_end, nullptr, (uint32_t)jb.prefix[sub * 4u], dc0, coef, jb.err);
}

} // namespace

size_t jh_blob_bytes(const JpegBlobHeader &H) { return (size_t)H.coef_off + (size_t)H.nblocks * 128u + 64u; }
uint32_t jh_subsequences(const JpegHuffStage &S) { return (S.stream_bits + kJhSubBits - 1u) / kJhSubBits; }

hipError_t launch_jpeg_huff(const JhJob *d_jobs, const JhJob *h_jobs, uint32_t njobs, const JhItem *d_items, uint32_t nitems, uint32_t max_blocks, hipStream_t st)
{
    (void)h_jobs;
    if (!njobs || !nitems) return hipSuccess;
    jh_init_kernel<<<dim3((max_blocks + 255u) / 256u, njobs), 256, 0, st>>>(d_jobs, max_blocks);
    constexpr uint32_t lds = kLdsWords * 4u;
    static bool attr_done[16] = {};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= 16 || !attr_done[dev]) { // (more than the 64 KB a kernel gets without asking; idempotent, so a race between two lanes only repeats it)
        const void *fns[3] = {reinterpret_cast<const void *>(&jh_sync_kernel<true>), reinterpret_cast<const void *>(&jh_sync_kernel<false>), reinterpret_cast<const void *>(&jh_write_kernel)};
        for (const void *f : fns)
            if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) attr_done[dev] = true;
    }
    jh_sync_kernel<true><<<nitems, 256, lds, st>>>(d_jobs, d_items);
    for (uint32_t r = 0; r < kJhSyncRounds; ++r) jh_sync_kernel<false><<<nitems, 256, lds, st>>>(d_jobs, d_items); // (each moves the chain across one more workgroup boundary)
    jh_scan_kernel<<<njobs, 256, 0, st>>>(d_jobs);
    jh_write_kernel<<<nitems, 256, lds, st>>>(d_jobs, d_items);
    return hipGetLastError();
}

} // namespace fl
