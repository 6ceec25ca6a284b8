// q3_gemv_inst.hip -- every instantiation of k_gemv the planner can select, in a translation unit of its own so that
// the library builds in parallel (make -j).  q3_engine.hip reaches the kernels only through gemv_pick() / find_cfg().
#include <hip/hip_runtime.h>
#include "q3_gemv.h"

using namespace q3;
typedef void (*GemvFn)(const GemvArgs);

namespace {

// tile shapes instantiated: group 64 (every listed model) gets the full set, other group sizes a
// single-row-run fallback (RU = 1, or 2 for SwiGLU).  Variants (k_gemv FIN / PF): FIN = DPP-chain fold of the group terms
// (group 64, rows a whole number of tiles), PF = second tile requested before the prologue (streaming launches).
// Instantiated: (FIN,PF) = (0,0) everywhere; (1,0) and (1,1) for the layer kernels; (0,1) for the classifier.
template <int PRO, int EPI, int LPG_T, int RU, int FIN = 0, int PF = 0>
GemvFn pick_ju(int JU) {
    if (JU == 1) return (GemvFn)k_gemv<PRO, EPI, LPG_T, RU, 1, FIN, PF>;
    if (JU == 2) { if constexpr (RU <= 4) return (GemvFn)k_gemv<PRO, EPI, LPG_T, RU, 2, FIN, PF>; }
    if (JU == 3) { if constexpr (RU <= 2) return (GemvFn)k_gemv<PRO, EPI, LPG_T, RU, 3, FIN, PF>; }
    if (JU == 4) { if constexpr (RU <= 2) return (GemvFn)k_gemv<PRO, EPI, LPG_T, RU, 4, FIN, PF>; }
    return nullptr;
}
template <int PRO, int EPI, int FIN, int PF>
GemvFn pick_ru64(int RU, int JU) {
    constexpr bool sw = (EPI == EPI_SWIGLU);
    if (RU == 8) return pick_ju<PRO, EPI, 4, 8, FIN, PF>(JU);
    if (RU == 4) return pick_ju<PRO, EPI, 4, 4, FIN, PF>(JU);
    if (RU == 2) return pick_ju<PRO, EPI, 4, 2, FIN, PF>(JU);
    if constexpr (!sw) { if (RU == 1) return pick_ju<PRO, EPI, 4, 1, FIN, PF>(JU); }
    return nullptr;
}
template <int PRO, int EPI>
GemvFn pick(int G, int RU, int JU, int FIN = 0, int PF = 0) {
    constexpr bool sw = (EPI == EPI_SWIGLU);
    if (G == 64) {
        if constexpr (EPI == EPI_LOGITS) {
            if (PF) return pick_ru64<PRO, EPI, 0, 1>(RU, JU);
            return pick_ru64<PRO, EPI, 0, 0>(RU, JU);
        } else {
            if (FIN && PF) return pick_ru64<PRO, EPI, 1, 1>(RU, JU);
            if (FIN) return pick_ru64<PRO, EPI, 1, 0>(RU, JU);
            return pick_ru64<PRO, EPI, 0, 0>(RU, JU);
        }
    }
    if constexpr (sw) { if (RU == 2) return pick_ju<PRO, EPI, 0, 2>(JU); }
    else { if (RU == 1) return pick_ju<PRO, EPI, 0, 1>(JU); }
    return nullptr;
}

}  // namespace

namespace q3inst {
struct GemvCfg { int pro, epi, n, wgt, ept, ru, ju, pf; GemvFn fn; };
}
using q3inst::GemvCfg;

namespace {
// ------------------------------------------------------------------------------------------------
// Shape-specialised GEMV launches (k_gemv with N_T > 0): every (prologue, epilogue, contraction length) of the listed
// models (0.6B dim 1024 / hidden 3072, 4B 2560 / 9728, 8B 4096 / 12288, all heads x head_dim 2048 / 4096)
// has one or more tile / workgroup-width candidates; the first entry of a role is the default, Q3_CFG_<FAMILY>=k picks the
// k-th (sweeps: tools/cfg_sweep.py), -1 forces the generic run-time-n kernel.  Anything not listed (test shapes,
// other group sizes) takes the generic path.
// ------------------------------------------------------------------------------------------------
#define Q3_CFG(PRO, EPI, N, WGT, EPT, RU, JU, PF) \
    {PRO, EPI, N, WGT, EPT, RU, JU, PF, (GemvFn)k_gemv<PRO, EPI, 4, RU, JU, Q3_CFG_FIN, PF, N, WGT, EPT>}
#define Q3_CFG_NORM_QKV(N, WGT, EPT, RU, JU, PF) Q3_CFG(PRO_NORM, EPI_QKV, N, WGT, EPT, RU, JU, PF), Q3_CFG(PRO_EMBED_NORM, EPI_QKV, N, WGT, EPT, RU, JU, PF)
// The first entry of a role is what the product runs.  The other candidates of a role (the forms that lost their A/B; they stay
// selectable through Q3_CFG_<FAMILY>=k for re-sweeps) exist in the developer build only: Q3_ALT(...).
#ifdef Q3_DEV
#define Q3_ALT(...) __VA_ARGS__,
#else
#define Q3_ALT(...)
#endif
// Notes on the entries of Q3_CFG_LIST (a macro: the list is expanded twice, reference-order fold and tolerance-mode tree fold):
// --- QKV: RMSNorm_att + quantize + wq|wk|wv
// (dim 1024, end of r03: the 512-thread / two-rows-per-wave forms lead the 1024-thread ones by 0.1 us per launch since the
// wave-0 block loads go through LDS; same-process A/B of the three 0.6B changes below: 1,552-1,567 -> 1,582-1,599 tok/s)
// (r04 sweep, 4B: 12 waves x 2 rows = 6,144 rows exactly, 6.18 vs 6.56 us; the same form at 4096 is slower, 8.33 vs 7.94 us)
// (r04) 12-wave workgroups: 6,144 rows = 3,072 waves x 2 rows, every byte of the launch requested at kernel entry
// --- W1|W3 + SwiGLU
// (r05 re-sweep on the 4B shape, six alternations on one box: 8-wave workgroups 1,367 vs 1,387 us per token at 32 tokens, 2,148 vs
// 2,164 at position 2,300 -- the 16-wave form had won in r04, before the attention and prologue changes of this round)
// (r04: a 12-wave PF = 1 form still spills 92 B at 4096 and ran at 30.4 us; 2560: 16.5 vs 12.6 us -- not kept)
// --- Wo behind the short-context attention kernel (xb arrives quantized): register-direct activation
// --- quantize + W2 (and Wo of the long-context plan)
// (r04: the whole 12 KiB row of a wave requested before the prologue, two 6 KiB tiles: 13.3 vs 11.8 us; 9728 as 2 x 5 KiB: 9.2
// vs 8.3 us -- request depth at entry is not what these launches wait for)
// --- final RMSNorm + classifier (streaming: two tiles requested before the prologue)
#define Q3_CFG_LIST \
    Q3_CFG_NORM_QKV(1024, 512, 2, 2, 1, 0), \
    Q3_ALT(Q3_CFG_NORM_QKV(1024, 1024, 1, 1, 1, 0), Q3_CFG_NORM_QKV(1024, 1024, 4, 1, 1, 0), Q3_CFG_NORM_QKV(1024, 256, 4, 2, 1, 0)) \
    Q3_CFG_NORM_QKV(2560, 768, 4, 2, 3, 0), \
    Q3_ALT(Q3_CFG_NORM_QKV(2560, 1024, 4, 1, 3, 0), Q3_CFG_NORM_QKV(2560, 1024, 4, 1, 3, 1), Q3_CFG_NORM_QKV(2560, 1024, 4, 2, 3, 0)) \
    Q3_CFG_NORM_QKV(4096, 1024, 4, 1, 4, 0), \
    Q3_ALT(Q3_CFG_NORM_QKV(4096, 1024, 4, 1, 4, 1), Q3_CFG_NORM_QKV(4096, 1024, 4, 2, 4, 0), Q3_CFG_NORM_QKV(4096, 768, 4, 2, 4, 0)) \
    Q3_CFG(PRO_NORM, EPI_SWIGLU, 1024, 512, 2, 4, 1, 0), \
    Q3_ALT(Q3_CFG(PRO_NORM, EPI_SWIGLU, 1024, 1024, 1, 2, 1, 0), Q3_CFG(PRO_NORM, EPI_SWIGLU, 1024, 1024, 4, 2, 1, 0), \
           Q3_CFG(PRO_NORM, EPI_SWIGLU, 1024, 256, 4, 4, 1, 0)) \
    Q3_CFG(PRO_NORM, EPI_SWIGLU, 2560, 512, 4, 2, 3, 0), \
    Q3_ALT(Q3_CFG(PRO_NORM, EPI_SWIGLU, 2560, 512, 4, 2, 3, 1), Q3_CFG(PRO_NORM, EPI_SWIGLU, 2560, 1024, 4, 2, 3, 0)) \
    Q3_CFG(PRO_NORM, EPI_SWIGLU, 4096, 1024, 4, 2, 4, 0), \
    Q3_ALT(Q3_CFG(PRO_NORM, EPI_SWIGLU, 4096, 512, 4, 2, 4, 1), Q3_CFG(PRO_NORM, EPI_SWIGLU, 4096, 512, 4, 2, 4, 0)) \
    Q3_CFG(PRO_PREQR, EPI_RESID, 2048, 256, 4, 1, 2, 0), \
    Q3_ALT(Q3_CFG(PRO_PREQR, EPI_RESID, 2048, 512, 4, 1, 2, 0)) \
    Q3_CFG(PRO_PREQR, EPI_RESID, 4096, 256, 4, 1, 4, 0), \
    Q3_ALT(Q3_CFG(PRO_PREQR, EPI_RESID, 4096, 256, 4, 2, 4, 0), Q3_CFG(PRO_PREQR, EPI_RESID, 4096, 512, 4, 1, 4, 0), \
           Q3_CFG(PRO_PREQR, EPI_RESID, 4096, 1024, 4, 1, 4, 0)) \
    Q3_CFG(PRO_QUANT, EPI_RESID, 3072, 256, 4, 1, 3, 0), \
    Q3_ALT(Q3_CFG(PRO_QUANT, EPI_RESID, 3072, 512, 4, 1, 3, 0), Q3_CFG(PRO_QUANT, EPI_RESID, 3072, 1024, 4, 1, 3, 0)) \
    Q3_CFG(PRO_QUANT, EPI_RESID, 9728, 1024, 4, 1, 2, 1), \
    Q3_ALT(Q3_CFG(PRO_QUANT, EPI_RESID, 9728, 512, 4, 1, 2, 1)) \
    Q3_CFG(PRO_QUANT, EPI_RESID, 12288, 1024, 4, 1, 4, 1), \
    Q3_ALT(Q3_CFG(PRO_QUANT, EPI_RESID, 12288, 512, 4, 1, 4, 1)) \
    Q3_CFG(PRO_QUANT, EPI_RESID, 2048, 512, 4, 1, 2, 0), \
    Q3_ALT(Q3_CFG(PRO_QUANT, EPI_RESID, 2048, 256, 4, 1, 2, 0)) \
    Q3_CFG(PRO_QUANT, EPI_RESID, 4096, 1024, 4, 1, 4, 0), \
    Q3_ALT(Q3_CFG(PRO_QUANT, EPI_RESID, 4096, 512, 4, 1, 4, 0)) \
    Q3_CFG(PRO_NORM, EPI_LOGITS, 1024, 512, 4, 8, 1, 1), \
    Q3_ALT(Q3_CFG(PRO_NORM, EPI_LOGITS, 1024, 256, 4, 8, 1, 1)) \
    Q3_CFG(PRO_NORM, EPI_LOGITS, 2560, 512, 4, 2, 3, 1), \
    Q3_ALT(Q3_CFG(PRO_NORM, EPI_LOGITS, 2560, 256, 4, 2, 3, 1)) \
    Q3_CFG(PRO_NORM, EPI_LOGITS, 4096, 512, 4, 2, 4, 1), \
    Q3_ALT(Q3_CFG(PRO_NORM, EPI_LOGITS, 4096, 256, 4, 2, 4, 1))
#define Q3_CFG_FIN 1
const GemvCfg kGemvCfgs[] = {
Q3_CFG_LIST
};
#undef Q3_CFG_FIN
// Q3_FLAG_FAST engines: the default form of every role with the wavefront-tree group fold (k_gemv FIN = 2); no alternates
#undef Q3_ALT
#define Q3_ALT(...)
#define Q3_CFG_FIN 2
const GemvCfg kGemvCfgsFast[] = {
Q3_CFG_LIST
};
#undef Q3_CFG_FIN
}  // namespace

namespace q3inst {
const GemvCfg* find_cfg(int pro, int epi, int n, int G, int which, bool fast) {
    if (G != 64 || which < 0) return nullptr;
    if (fast) {                      // tolerance mode: the default form of the role with the tree fold (candidate 0 only)
        if (which != 0) return nullptr;
        for (const GemvCfg& c : kGemvCfgsFast)
            if (c.pro == pro && c.epi == epi && c.n == n) return &c;
        return nullptr;
    }
    for (const GemvCfg& c : kGemvCfgs)
        if (c.pro == pro && c.epi == epi && c.n == n && which-- == 0) return &c;
    return nullptr;                  // no such candidate: the caller falls back to the generic kernel (a sweep sees "generic", not a mislabel)
}


// the (prologue, epilogue) pairs the engine and the operator entry points use
GemvFn gemv_pick(int pro, int epi, int G, int RU, int JU, int FIN, int PF) {
    if (pro == PRO_EMBED_NORM && epi == EPI_QKV) return pick<PRO_EMBED_NORM, EPI_QKV>(G, RU, JU, FIN, PF);
    if (pro == PRO_NORM && epi == EPI_QKV) return pick<PRO_NORM, EPI_QKV>(G, RU, JU, FIN, PF);
    if (pro == PRO_QUANT && epi == EPI_RESID) return pick<PRO_QUANT, EPI_RESID>(G, RU, JU, FIN, PF);
    if (pro == PRO_NORM && epi == EPI_SWIGLU) return pick<PRO_NORM, EPI_SWIGLU>(G, RU, JU, FIN, PF);
    if (pro == PRO_NORM && epi == EPI_LOGITS) return pick<PRO_NORM, EPI_LOGITS>(G, RU, JU, FIN, PF);
    if (pro == PRO_PREQ && epi == EPI_STORE) return pick<PRO_PREQ, EPI_STORE>(G, RU, JU, FIN, PF);
    return nullptr;
}
}  // namespace q3inst
