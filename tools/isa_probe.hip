// isa_probe.hip -- diagnostic only (never linked into the library): one tiny kernel per hot routine of ugs_kernels.hip, so that
// their instruction counts can be read off the ISA (tools/isa_probe.py).  Compile: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only.
#include "../ss-gnn_amd/csrc/ugs_kernels.hip"

namespace {
struct ProbeWs {
    Work<LdsSpace> ws;
    uint32_t *SV;
    uint4 *EL;
    __device__ ProbeWs(uint32_t *lds) {
        using Cfg = TierCfg<448>;
        ws.D = lds; ws.ORD = reinterpret_cast<uint16_t *>(ws.D + 448); ws.AUX = nullptr; ws.TBL = ws.D + 448 + Cfg::ORDW; ws.HK = ws.TBL + Cfg::BCAP_A;
        SV = ws.HK + Cfg::HS; EL = reinterpret_cast<uint4 *>(SV + UGS_KMAX);
        ws.cap = 448; ws.hmask = Cfg::HS - 1; ws.hlimit = Cfg::HLIMIT;
    }
};
#define PROBE_PRE __shared__ __attribute__((aligned(16))) uint32_t lds[TierCfg<448>::WORDS]; Grp<64> g; g.init(); ProbeWs pw(lds); lds[threadIdx.x] = in[threadIdx.x]; __syncthreads();

__global__ __launch_bounds__(64, 5) void probe_empty(uint32_t *out, const uint32_t *in, uint32_t c, uint32_t rsel) {
    PROBE_PRE
    out[threadIdx.x] = lds[(threadIdx.x + c) & 63] + rsel;
}
template <int STAGE> __global__ __launch_bounds__(64, 5) void probe_mat(uint32_t *out, const uint32_t *in, uint32_t c, uint32_t rsel) {
    PROBE_PRE
    mat_at<64, 7, STAGE>(pw.ws, g);
    out[threadIdx.x] = lds[(threadIdx.x + c) & 63] + rsel;
}
template <int STAGE, int NJ> __global__ __launch_bounds__(64, 5) void probe_matlds(uint32_t *out, const uint32_t *in, uint32_t c, uint32_t rsel) {
    PROBE_PRE
    using C = ChainAt<STAGE>;
    stage_mat<64, NJ>(pw.ws, g, pw.ws.ORD + C::OOLD, pw.ws.ORD + C::O, C::NOLD, C::B, C::M, C::S);
    out[threadIdx.x] = lds[(threadIdx.x + c) & 63] + rsel;
}
template <int STAGE, int NJ, bool REG = true> __global__ __launch_bounds__(64, 5) void probe_final(uint32_t *out, const uint32_t *in, uint32_t c, uint32_t rsel) {
    PROBE_PRE
    using C = ChainAt<STAGE>;
    Pick p;
    if constexpr (!REG) p = stage_final<64, NJ>(pw.ws, g, pw.ws.ORD + C::OOLD, C::NOLD, c, C::B, C::M, C::S, rsel);
    else if constexpr (NJ == 1) p = stage_final_reg(pw.ws, g, pw.ws.ORD + C::OOLD, C::NOLD, c, C::B, C::M, C::S, rsel);
    else p = stage_final<64, NJ>(pw.ws, g, pw.ws.ORD + C::OOLD, C::NOLD, c, C::B, C::M, C::S, rsel);
    out[threadIdx.x] = lds[(threadIdx.x + c) & 63] + p.w + p.q;
}
template <bool ADD> __global__ __launch_bounds__(64, 5) void probe_chunk(uint32_t *out, const uint32_t *in, uint32_t c, uint32_t rsel, const int2 *adj, uint2 *stage) {
    PROBE_PRE
    StageCtx sc; sc.EL = pw.EL; sc.ne = rsel & 3; sc.on = stage != nullptr; { uint32_t on32 = (uint32_t)__builtin_amdgcn_readfirstlane(sc.on ? 1 : 0); asm volatile("" : "+s"(on32)); sc.onm = 0ull - (uint64_t)on32; }
    uint32_t hcount = c + 3, ecount = rsel >> 8, cc = c;
    int2 e = UGS_NO_ENTRY; if (threadIdx.x < 40) e = adj[threadIdx.x];
    const bool ok = scan_chunk<64, LdsSpace, ADD, true>(pw.ws, g, rsel, c >> 3, 3, cc, hcount, ecount, sc, e, threadIdx.x + 1000u);
    out[threadIdx.x] = lds[(threadIdx.x + c) & 63] + cc + hcount + ecount + sc.ne + (ok ? 1u : 0u);
}
__global__ __launch_bounds__(64, 5) void probe_draw(uint32_t *out, const uint32_t *in, uint32_t c, uint32_t rsel) {
    PROBE_PRE
    Rng rng; rng.init(((uint64_t)rsel << 32) | c);
    const uint32_t r = g.uni(mod64_by<true>(rng.next(), c));
    out[threadIdx.x] = lds[(threadIdx.x + c) & 63] + r + (uint32_t)rng.s;
}
__global__ __launch_bounds__(64, 5) void probe_flush(uint32_t *out, const uint32_t *in, uint32_t c, uint32_t rsel, const int2 *adjf, uint2 *stage) {
    PROBE_PRE
    uint4 en = make_uint4(0, 0, 0, 0); uint32_t ecol = 0;
    if (threadIdx.x < c) { en = pw.EL[threadIdx.x]; ecol = (uint32_t)adjf[en.x].y; }
    stage_flush(c, g, pw.SV, rsel, en, ecol, stage);
    out[threadIdx.x] = lds[(threadIdx.x + c) & 63];
}
template __global__ void probe_mat<0>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_mat<1>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_mat<2>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_mat<3>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_mat<4>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<2, 1>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<4, 3>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<4, 5>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<5, 5>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<5, 7>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<2, 1, false>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<3, 2, false>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<4, 4, false>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_final<5, 6, false>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_matlds<2, 1>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_matlds<3, 2>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_matlds<3, 3>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_matlds<4, 5>(uint32_t *, const uint32_t *, uint32_t, uint32_t);
template __global__ void probe_chunk<true>(uint32_t *, const uint32_t *, uint32_t, uint32_t, const int2 *, uint2 *);
template __global__ void probe_chunk<false>(uint32_t *, const uint32_t *, uint32_t, uint32_t, const int2 *, uint2 *);
}  // namespace
