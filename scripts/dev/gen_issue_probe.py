#!/usr/bin/env python3
"""Dev tool: generate scripts/dev/bin/issue_probe.hip -- instruction-issue probes for the delay-and-sum sweep on gfx950.

Each probe kernel is the inner loop of a candidate sweep design written as ONE asm block with hard-wired registers:
per microphone a table fetch (s_load), the quads of F frames (LDS reads, in place or prefetched one mic ahead), and DW
direction steps of 4 F packed operations (lerp: 2 v_pk_fma + 2 v_pk_add per frame) separated by the offset test
(s_cmp + not-taken s_cbranch, reload stub out of line).  What is measured is the time per packed operation against a
stream of nothing but packed operations at the same occupancy -- i.e. what every non-arithmetic instruction of a design
costs, before the real kernel is written.

  python3 scripts/dev/gen_issue_probe.py && hipcc --offload-arch=gfx950 -O3 scripts/dev/bin/issue_probe.hip -o scripts/dev/bin/issue_probe
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# name, waves per CU, F frames, DW directions, read kind, prefetch, checks, lerp, table loads
#   read kind: "b64c" two ds_read_b64 at 16-byte lane stride (2-way bank conflict, today's kernel), "b64" the same at 8-byte lane
#   stride (conflict-free), "r2" one ds_read2_b64 (offset1 = 64 -> +512 bytes), "none" no LDS reads
CONFIGS = [
    # calibration: packed only
    dict(name="pure16", waves=16, F=2, DW=8, read="none", pre=False, check=False, lerp=True, tab=False),
    dict(name="pure8", waves=8, F=4, DW=8, read="none", pre=False, check=False, lerp=True, tab=False),
    # one ingredient at a time on the F=2, DW=8, 16-wave shape (today's pair kernel)
    dict(name="chk16", waves=16, F=2, DW=8, read="none", pre=False, check=True, lerp=True, tab=False),
    dict(name="tab16", waves=16, F=2, DW=8, read="none", pre=False, check=False, lerp=True, tab=True),
    dict(name="rdc16", waves=16, F=2, DW=8, read="b64c", pre=False, check=False, lerp=True, tab=False),
    dict(name="rdf16", waves=16, F=2, DW=8, read="b64", pre=False, check=False, lerp=True, tab=False),
    dict(name="rd2_16", waves=16, F=2, DW=8, read="r2", pre=False, check=False, lerp=True, tab=False),
    # whole designs
    dict(name="cur16", waves=16, F=2, DW=8, read="b64c", pre=False, check=True, lerp=True, tab=True),    # today's pair kernel
    dict(name="cur16_r2", waves=16, F=2, DW=8, read="r2", pre=False, check=True, lerp=True, tab=True),   # + conflict-free read2
    dict(name="x16_f4d4", waves=16, F=4, DW=4, read="r2", pre=False, check=True, lerp=True, tab=True),   # 4 frames x 4 directions
    dict(name="y8_f4d8", waves=8, F=4, DW=8, read="r2", pre=False, check=True, lerp=True, tab=True),     # 4 frames x 8 directions, 256 VGPRs
    dict(name="y8_f4d8_pre", waves=8, F=4, DW=8, read="r2", pre=True, check=True, lerp=True, tab=True),  # ... quads prefetched a mic ahead
    # today's shape with the next mic's first quads requested a mic ahead (two quad sets, hard-wired registers)
    dict(name="pre16_c", waves=16, F=2, DW=8, read="b64c", pre=True, check=True, lerp=True, tab=True),
    dict(name="pre16_f", waves=16, F=2, DW=8, read="b64", pre=True, check=True, lerp=True, tab=True),
    dict(name="pre16_r2", waves=16, F=2, DW=8, read="r2", pre=True, check=True, lerp=True, tab=True),
    # ... and re-reads that land in a spare quad set while the step's arithmetic runs, then 8 v_mov_b64
    dict(name="alt16_c", waves=16, F=2, DW=8, read="b64c", pre=True, check=True, lerp=True, tab=True, alt=True),
    dict(name="alt16_f", waves=16, F=2, DW=8, read="b64", pre=True, check=True, lerp=True, tab=True, alt=True),
    dict(name="pad_pre16_c", waves=16, F=2, DW=8, read="b64c", pre=True, check=True, lerp=False, tab=True),
    dict(name="pad_alt16_c", waves=16, F=2, DW=8, read="b64c", pre=True, check=True, lerp=False, tab=True, alt=True),
    # frames interleaved in LDS: 4 ds_read_b128 per (re)load instead of 8 ds_read_b64
    dict(name="cur16_b128", waves=16, F=2, DW=8, read="b128", pre=False, check=True, lerp=True, tab=True),
    dict(name="pre16_b128", waves=16, F=2, DW=8, read="b128", pre=True, check=True, lerp=True, tab=True),
    dict(name="pad_cur16_b128", waves=16, F=2, DW=8, read="b128", pre=False, check=True, lerp=False, tab=True),
    # pad flavours
    dict(name="pad_pure16", waves=16, F=2, DW=8, read="none", pre=False, check=False, lerp=False, tab=False),
    dict(name="pad_cur16", waves=16, F=2, DW=8, read="b64c", pre=False, check=True, lerp=False, tab=True),
    dict(name="pad_cur16_r2", waves=16, F=2, DW=8, read="r2", pre=False, check=True, lerp=False, tab=True),
    dict(name="pad_y8_f4d8_pre", waves=8, F=4, DW=8, read="r2", pre=True, check=True, lerp=False, tab=True),
    dict(name="pad_y8_f8d8_pre", waves=8, F=8, DW=4, read="r2", pre=True, check=True, lerp=False, tab=True),
]


def gen_kernel(c):
    F, DW, lerp = c["F"], c["DW"], c["lerp"]
    A = 2 if lerp else 1                       # arrays per frame: S (and D)
    # VGPR map: v0 = lane base (set in the prologue below), v1.. = address temporaries.  ds_read2_b64 has 8-bit offsets in units
    # of 8 bytes, i.e. one address reaches two consecutive 1248-byte rows: one address per two (frame, array) rows.
    nrows = F * A
    naddr = (nrows + 1) // 2 if c["read"] == "r2" else 1
    acc0 = 2 * ((1 + naddr + 1) // 2)          # accumulators: DW x F x 4 (64-bit operands sit on even registers)
    nacc = DW * F * 4
    q0 = acc0 + nacc                           # quad sets: [set][frame][array][4]
    nq = F * A * 4
    alt = bool(c.get("alt"))
    sets = (2 if c["pre"] else 1) + (1 if alt else 0)
    ALT = sets - 1
    t0 = q0 + sets * nq                        # product temporaries: F x 4
    nt = F * 4 if lerp else 0
    top = t0 + nt
    assert top <= (256 if c["waves"] == 8 else 128), (c["name"], top)

    def acc(j, f, h):
        b = acc0 + ((j * F + f) * 2 + h) * 2
        return "v[%d:%d]" % (b, b + 1)

    def quad(s, f, a, h):
        b = q0 + s * nq + ((f * A + a) * 2 + h) * 2
        return "v[%d:%d]" % (b, b + 1)

    def quad4(s, f, a):
        b = q0 + s * nq + (f * A + a) * 4
        return "v[%d:%d]" % (b, b + 3)

    def tmp(f, h):
        b = t0 + (f * 2 + h) * 2
        return "v[%d:%d]" % (b, b + 1)

    # SGPR map (hard-wired, above what the compiler needs for the few C++ values): s40.. offsets, s56.. weights, two sets
    def e(p, j):
        return "s%d" % (40 + p * 16 + j)

    def hp(p, j):
        b = 40 + p * 16 + 8 + (j // 2) * 2
        return "s[%d:%d]" % (b, b + 1)

    rowb = 1248                                # bytes per staged row (312 floats), rows of a mic back to back
    L = []

    def addrs(ev):
        if c["read"] == "r2":
            for i in range(naddr):
                L.append("v_add_u32 v%d, %s, v0" % (1 + i, ev) if i == 0 else "v_add_u32 v%d, %d, v1" % (1 + i, 2 * rowb * i))
        elif c["read"] != "none":
            L.append("v_add_u32 v1, %s, v0" % ev)

    def reads(s):
        if c["read"] == "b128":
            # frames interleaved sample by sample: per array two 16-byte reads (sample pairs 2l, 2l+1 and 128+2l, 128+2l+1 of both frames)
            assert F == 2
            for a in range(A):
                for hh in range(2):
                    b = q0 + s * nq + (a * 2 + hh) * 4
                    L.append("ds_read_b128 v[%d:%d], v1 offset:%d" % (b, b + 3, a * 2 * rowb + hh * 1024))
            return
        for f in range(F):
            for a in range(A):
                r = f * A + a
                if c["read"] == "r2":
                    off = (r % 2) * rowb
                    L.append("ds_read2_b64 %s, v%d offset0:%d offset1:%d" % (quad4(s, f, a), 1 + r // 2, off // 8, off // 8 + 64))
                elif c["read"] == "b128":
                    pass
                elif c["read"] in ("b64", "b64c"):
                    off = r * rowb
                    second = 512 if c["read"] == "b64" else 8
                    L.append("ds_read_b64 %s, v1 offset:%d" % (quad(s, f, a, 0), off))
                    L.append("ds_read_b64 %s, v1 offset:%d" % (quad(s, f, a, 1), off + second))

    def step(s, p, j):
        if lerp:
            mods = "op_sel_hi:[0,1,1]" if j % 2 == 0 else "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
            for f in range(F):
                for h in range(2):
                    L.append("v_pk_fma_f32 %s, %s, %s, %s %s" % (tmp(f, h), hp(p, j), quad(s, f, 1, h), quad(s, f, 0, h), mods))
            for f in range(F):
                for h in range(2):
                    L.append("v_pk_add_f32 %s, %s, %s" % (acc(j, f, h), acc(j, f, h), tmp(f, h)))
        else:
            for f in range(F):
                for h in range(2):
                    L.append("v_pk_add_f32 %s, %s, %s" % (acc(j, f, h), acc(j, f, h), quad(s, f, 0, h)))

    stubs = []

    def mic(p, uid):
        s = p if c["pre"] else 0
        L.append("s_waitcnt lgkmcnt(0)")
        if c["read"] != "none":
            if c["pre"]:
                addrs(e(p, 1))
                reads(s ^ 1)                   # next mic's quads
            else:
                addrs(e(p, 0))
                reads(s)
                L.append("s_waitcnt lgkmcnt(0)")
        for j in range(DW):
            if j == 1 and c["tab"]:
                # the next mic's table rows (after the in-place read's wait, as the real kernel does)
                L.append("s_load_dwordx8 s[%d:%d], s[2:3], 0x0" % (40 + (p ^ 1) * 16, 40 + (p ^ 1) * 16 + 7))
                L.append("s_load_dwordx8 s[%d:%d], s[2:3], 0x20" % (40 + (p ^ 1) * 16 + 8, 40 + (p ^ 1) * 16 + 15))
                L.append("s_add_u32 s2, s2, 64")
                L.append("s_addc_u32 s3, s3, 0")
            if alt:
                if j < DW - 1 and c["check"]:
                    # test the NEXT step's offset before this step's arithmetic; a re-read lands in the spare set meanwhile
                    lab = "%d_%d" % (uid, j)
                    L.append("s_cmp_lg_u32 %s, %s" % (e(p, j + 1), e(p, j)))
                    L.append("s_cbranch_scc1 .Lr%s_%%=" % lab)
                    keep = L[:]
                    del L[:]
                    L.append(".Lr%s_%%=:" % lab)
                    addrs(e(p, j + 1))
                    reads(ALT)
                    step(s, p, j)
                    L.append("s_waitcnt lgkmcnt(0)")
                    for r in range(0, nq, 2):
                        L.append("v_mov_b64 v[%d:%d], v[%d:%d]" % (q0 + s * nq + r, q0 + s * nq + r + 1, q0 + ALT * nq + r, q0 + ALT * nq + r + 1))
                    L.append("s_branch .Lb%s_%%=" % lab)
                    stubs.extend(L)
                    del L[:]
                    L.extend(keep)
                    step(s, p, j)
                    L.append(".Lb%s_%%=:" % lab)
                else:
                    step(s, p, j)
                continue
            if j > 0 and c["check"]:
                lab = "%d_%d" % (uid, j)
                L.append("s_cmp_lg_u32 %s, %s" % (e(p, j), e(p, j - 1)))
                L.append("s_cbranch_scc1 .Lr%s_%%=" % lab)
                L.append(".Lb%s_%%=:" % lab)
                st = [".Lr%s_%%=:" % lab]
                keep = L[:]
                del L[:]
                addrs(e(p, j))
                reads(s)
                st += L[:]
                del L[:]
                L.extend(keep)
                st += ["s_waitcnt lgkmcnt(0)", "s_branch .Lb%s_%%=" % lab]
                stubs.extend(st)
            step(s, p, j)

    # prologue: zero everything, lane base, first table rows
    L.append("v_mbcnt_lo_u32_b32 v0, -1, 0")
    L.append("v_mbcnt_hi_u32_b32 v0, -1, v0")
    L.append("v_lshlrev_b32 v0, %d, v0" % (4 if c["read"] in ("b64c", "b128") else 3))
    for r in range(acc0, top):
        L.append("v_mov_b32 v%d, 0" % r)
    L.append("s_mov_b64 s[2:3], %[tab]")
    L.append("s_mov_b32 s4, %[mics]")
    L.append("s_load_dwordx8 s[40:47], s[2:3], 0x0")
    L.append("s_load_dwordx8 s[48:55], s[2:3], 0x20")
    L.append("s_load_dwordx8 s[56:63], s[2:3], 0x0")
    L.append("s_load_dwordx8 s[64:71], s[2:3], 0x20")
    L.append("s_waitcnt lgkmcnt(0)")
    L.append(".Lloop_%=:")
    mic(0, 0)
    mic(1, 1)
    L.append("s_sub_u32 s4, s4, 2")
    L.append("s_cmp_lg_u32 s4, 0")
    L.append("s_cbranch_scc1 .Lloop_%=")
    L.append("s_waitcnt lgkmcnt(0)")
    # fold the accumulators into one value so that the stores below depend on everything
    for r in range(acc0 + 1, acc0 + nacc):
        L.append("v_add_f32 v%d, v%d, v%d" % (acc0, acc0, r))
    L.append("v_mov_b32 %[res], v" + str(acc0))
    L.append("s_branch .Lend_%=")
    L.extend(stubs)
    L.append(".Lend_%=:")
    clob = ", ".join('"v%d"' % r for r in range(0, top)) + ", " + ", ".join('"s%d"' % r for r in range(2, 5)) + ", " + \
        ", ".join('"s%d"' % r for r in range(40, 72)) + ', "scc", "memory"'
    body = "\n".join('        "%s\\n\\t"' % x for x in L)
    packed_per_mic = DW * F * (4 if lerp else 2)
    return """
__global__ void __launch_bounds__(%d) k_%s(const int* __restrict__ tab, float* __restrict__ out, int mics)
{
    extern __shared__ float lds[];
    if (threadIdx.x == 100000) lds[threadIdx.x] = 0.f;
    float res;
    asm volatile(
%s
        : [res] "=v"(res) : [tab] "s"(tab), [mics] "s"(mics) : %s);
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
}
""" % (c["waves"] * 64, c["name"], body, clob), packed_per_mic


def main():
    parts = ["// generated by scripts/dev/gen_issue_probe.py -- do not edit\n#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>\n#include <vector>\n"
             "#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf(\"%s: %s\\n\", #x, hipGetErrorString(e_)); return 1; } } while (0)\n"]
    runs = []
    for c in CONFIGS:
        k, ppm = gen_kernel(c)
        parts.append(k)
        runs.append((c, ppm))
    parts.append("""
int main(int argc, char** argv)
{
    const int mics = 4096, blocks = 256 * 4;
    const char* only = argc > 1 && strcmp(argv[1], "all") ? argv[1] : nullptr;
    const double p_change = argc > 2 ? atof(argv[2]) : 0.0;   // probability that a direction step changes the LDS offset
    int* d_tab; float* d_out;
    std::vector<int> tab(16 * (mics + 8), 0);
    unsigned long long rng = 88172645463325252ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; };
    long long changes = 0;
    for (int m = 0; m < mics + 8; ++m) {
        int ecur = 16 * (int)(rnd() * 8);
        for (int j = 0; j < 8; ++j) {
            if (j > 0 && rnd() < p_change) { ecur = (ecur + 16) % 256; ++changes; }
            tab[16 * m + j] = ecur;
        }
        for (int j = 8; j < 16; ++j) { float h = 0.25f + 0.001f * j; memcpy(&tab[16 * m + j], &h, 4); }
    }
    printf("offset changes per mic: %.3f\\n", (double)changes / (mics + 8));
    CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)blocks * 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double base16 = 0, base8 = 0, pbase16 = 0;
""")
    for c, ppm in runs:
        parts.append("""
    if (!only || !strcmp(only, "%(name)s")) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_%(name)s), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_%(name)s, dim3(blocks), dim3(%(threads)d), 160 * 1024, 0, d_tab, d_out, mics);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        // packed operations per SIMD: blocks/256 rounds x waves/4 waves x mics x packed per mic
        const double packed = (double)(blocks / 256) * (%(waves)d / 4.0) * mics * %(ppm)d;
        const double ns = best * 1e6 / packed;
        if (!strcmp("%(name)s", "pure16")) base16 = ns;
        if (!strcmp("%(name)s", "pure8")) base8 = ns;
        if (!strcmp("%(name)s", "pad_pure16")) pbase16 = ns;
        const double base = %(lerp)d ? (%(waves)d == 16 ? base16 : base8) : pbase16;
        printf("%%-18s waves %%2d F %%d DW %%d: %%8.3f ms  %%.3f ns per packed op per SIMD  (x%%.3f of packed-only)\\n", "%(name)s", %(waves)d, %(F)d, %(DW)d, best, ns, base > 0 ? ns / base : 0.0);
    }
""" % dict(name=c["name"], threads=c["waves"] * 64, waves=c["waves"], ppm=ppm, F=c["F"], DW=c["DW"], lerp=1 if c["lerp"] else 0))
    parts.append("    return 0;\n}\n")
    out = os.path.join(ROOT, "scripts", "dev", "bin", "issue_probe.hip")   # (gpurun_out/ does not travel to the GPU box)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write("".join(parts))
    print("wrote", out)


if __name__ == "__main__":
    main()
