#!/usr/bin/env python3
"""Generator of tools/micro/mfma_fillers.hip: what does one `v_mfma_f32_32x32x16_bf16` GAP cost with k fillers in it?

VERDICT r4 item 1: settle the issue model of one gfx950 SIMD before rebuilding the attention backward.  Every kernel below is ONE
asm statement on hard-coded registers (declared as clobbers, so hipcc neither schedules nor pads anything inside): 16 or 20 gaps
per loop iteration, each gap = one MFMA followed by k fillers of one kind (or a hand-written 10-gap block of the attention
backward's real mix).  Timed with s_memtime (shader cycles; s_memrealtime beside it gives the clock) at one wave per SIMD
(256-thread workgroups) and at two (512 threads), one workgroup per CU (100 KiB of LDS each).

    python3 tools/micro/gen_mfma_fillers.py > tools/micro/mfma_fillers.hip
    hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_fillers tools/micro/mfma_fillers.hip      (on the GPU box: ./mfma_fillers)

Register map (per lane): v[0:15] compiler; accumulators v[32:47], v[48:63], v[64:79], v[80:95]; A v[96:99], B v[100:103];
sources v104..v111; filler destinations v[112:175] (rotating); LDS address v176; LDS data v[178:179].
"""
import sys

ACC = [32, 48, 64, 80]
A, B = 96, 100
SRC = 104
DST = 112
NDST = 64
ADDR = 176
LDAT = 178


class Body:
    def __init__(self):
        self.lines = []
        self.rot = 0
        self.lds_off = 0

    def dst(self, n=1):
        # n consecutive destination registers, aligned to n
        r = self.rot
        if r % n:
            r += n - r % n
        if r + n > NDST:
            r = 0
        self.rot = r + n
        return DST + r

    def mfma(self, acc=None, dep=True):
        a = ACC[len([l for l in self.lines if 'v_mfma' in l]) % 4] if acc is None else acc
        self.lines.append(f'v_mfma_f32_32x32x16_bf16 v[{a}:{a + 15}], v[{A}:{A + 3}], v[{B}:{B + 3}], v[{a}:{a + 15}]')

    def filler(self, kind):
        if kind == 'exp':
            self.lines.append(f'v_exp_f32 v{self.dst()}, v{SRC}')
        elif kind == 'cvt':
            self.lines.append(f'v_cvt_pk_bf16_f32 v{self.dst()}, v{SRC + 1}, v{SRC + 2}')
        elif kind == 'mul':
            self.lines.append(f'v_mul_f32 v{self.dst()}, v{SRC + 1}, v{SRC + 2}')
        elif kind == 'add':
            self.lines.append(f'v_add_f32 v{self.dst()}, v{SRC + 1}, v{SRC + 2}')
        elif kind == 'pkmul':
            d = self.dst(2)
            self.lines.append(f'v_pk_mul_f32 v[{d}:{d + 1}], v[{SRC + 2}:{SRC + 3}], v[{SRC + 4}:{SRC + 5}]')
        elif kind == 'dsw':      # ds_write_b64, conflict free (lane * 8)
            self.lines.append(f'ds_write_b64 v{ADDR}, v[{LDAT}:{LDAT + 1}] offset:{self.next_off()}')
        elif kind == 'dsw128':
            self.lines.append(f'ds_write_b128 v{ADDR + 1}, v[{SRC}:{SRC + 3}] offset:{self.next_off(1024)}')
        elif kind == 'dsr_tr':   # ds_read_b64_tr_b16
            d = self.dst(2)
            self.lines.append(f'ds_read_b64_tr_b16 v[{d}:{d + 1}], v{ADDR} offset:{self.next_off()}')
        elif kind == 'dsr128':
            d = self.dst(4)
            self.lines.append(f'ds_read_b128 v[{d}:{d + 3}], v{ADDR + 1} offset:{self.next_off(1024)}')
        elif kind == 'nop':
            self.lines.append('s_nop 0')
        elif kind == 'mov':
            self.lines.append(f'v_mov_b32 v{self.dst()}, v{SRC + 1}')
        else:
            raise ValueError(kind)

    def next_off(self, step=512):
        o = self.lds_off
        self.lds_off = (self.lds_off + step) % 32768
        return o

    def text(self):
        return '\\n\\t'.join(self.lines)


def uniform(kind, k, gaps=16):
    b = Body()
    for _ in range(gaps):
        b.mfma()
        for _ in range(k):
            b.filler(kind)
    return b, gaps


# the product kernel's block (svol_amd/csrc/attention_bf16.hip, slots 1..10 of attn_bwd_sp_body), fillers by kind; the per-step
# skeleton is left out.  E exp, C cvt_pk, P pk_mul, M mul, W ds_write_b64, R ds_read_b64_tr_b16, L ds_read_b128
PROD = ['EEEE', 'CCEEP', 'EECC', 'PEECCP', 'EECP', 'CCECP', 'EEEC', 'CPPPCC', 'CCWW', 'WWRRRR']
# pk_mul -> two plain multiplies, same places
PROD_MUL = [s.replace('P', 'MM') for s in PROD]
# balanced by issue cost (exp 8, others ~4-6): 16 E, 16 C, 16 M, 4 W, 4 R over 10 gaps, at most two exps per gap
def balanced(counts, gaps=10):
    cost = {'E': 8.0, 'C': 4.5, 'M': 4.0, 'W': 6.0, 'R': 4.0, 'P': 8.0}
    slots = [[] for _ in range(gaps)]
    load = [0.0] * gaps
    for ch in sorted(counts, key=lambda c: -cost[c]):
        for _ in range(counts[ch]):
            # lightest gap; exps at most two per gap
            order = sorted(range(gaps), key=lambda g: (load[g], g))
            for g in order:
                if ch == 'E' and slots[g].count('E') >= 2:
                    continue
                slots[g].append(ch)
                load[g] += cost[ch]
                break
    out = []
    for sl in slots:   # interleave: an exp, then cheap ones, then the other exp
        es = [c for c in sl if c == 'E']
        rest = [c for c in sl if c != 'E']
        seq = []
        if es:
            seq.append(es.pop())
        half = len(rest) // 2
        seq += rest[:half]
        if es:
            seq.append(es.pop())
        seq += rest[half:]
        out.append(''.join(seq))
    return out


BAL = balanced({'E': 16, 'C': 16, 'M': 16, 'W': 4, 'R': 4})
# the same multiset with the exps in clusters of four (what a compiler would emit)
CLUMP = ['EEEE', 'EEEE', 'EEEE', 'EEEE', 'MMMMMMMM', 'MMMMMMMM', 'CCCCCCCC', 'CCCCCCCC', 'WWWW', 'RRRR']
# the minimal block plus the per-step skeleton of an 8-wave (2 blocks per step) kernel spread over its 20 gaps: 12 L (operands,
# row constants), 8 R (transposed A operands), 4 dsw128 (partial), 8 dsr (reduce, as R), 14 adds, 2 nops (atomics stand-ins)
KINDS = {'E': 'exp', 'C': 'cvt', 'P': 'pkmul', 'M': 'mul', 'W': 'dsw', 'R': 'dsr_tr', 'L': 'dsr128', 'A': 'add', 'N': 'nop', 'S': 'dsw128', 'V': 'mov'}


def block(slots, reps=2):
    b = Body()
    for _ in range(reps):
        for s in slots:
            b.mfma()
            for ch in s:
                b.filler(KINDS[ch])
    return b, len(slots) * reps


def count(slots):
    from collections import Counter
    return dict(Counter(''.join(slots)))


VARIANTS = []
for kind in ['exp', 'cvt', 'mul', 'pkmul', 'dsw', 'dsr_tr', 'dsr128', 'nop', 'mov']:
    for k in range(0, 11):
        if kind != 'exp' and k == 0:
            continue
        VARIANTS.append((f'{kind}_{k}', f'{k} x {kind} per gap') + uniform(kind, k))
VARIANTS.append(('blk_prod', 'product block (pk_mul), 10 gaps: ' + ' '.join(PROD)) + block(PROD))
VARIANTS.append(('blk_prod_mul', 'product block, pk_mul -> 2 mul: ' + ' '.join(PROD_MUL)) + block(PROD_MUL))
VARIANTS.append(('blk_bal', 'balanced block: ' + ' '.join(BAL)) + block(BAL))
VARIANTS.append(('blk_clump', 'clustered block: ' + ' '.join(CLUMP)) + block(CLUMP))
# only the arithmetic (no LDS): 16 E 16 C 16 M
ARITH = ['EEMC', 'EEMC', 'EMMCC', 'EEMC', 'EMMCC', 'EEMC', 'EMMC', 'EEMC', 'EMMCC', 'EEMC']
VARIANTS.append(('blk_arith', 'arithmetic only: ' + ' '.join(ARITH)) + block(ARITH))
ARITH_PK = ['EECP', 'EECP', 'ECCP', 'EECP', 'ECCP', 'EECP', 'ECCP', 'EECP', 'ECC', 'EEC']
VARIANTS.append(('blk_arith_pk', 'arithmetic only, 8 pk_mul: ' + ' '.join(ARITH_PK)) + block(ARITH_PK))
# 8-wave step: two balanced blocks + the per-step skeleton (12 L, 8 R, 4 S, 8 R as the reduce reads, 14 A, 2 N) over 20 gaps
STEP8 = list(BAL) + list(BAL)
extra = list('LLLLLLLLLLLL' 'RRRRRRRR' 'SSSS' 'RRRRRRRR' 'AAAAAAAAAAAAAA' 'NN')
for i, ch in enumerate(extra):
    STEP8[(i * 7) % 20] += ch
VARIANTS.append(('step8', '8-wave step (2 blocks + skeleton): ' + ' '.join(STEP8)) + block(STEP8, reps=1))
STEP4 = list(BAL) * 4
extra4 = list('LLLLLLLLLLLL' 'RRRRRRRR' 'SSSS' 'LLLL' 'AAAAAAAAAAAA' 'NNNN')
for i, ch in enumerate(extra4):
    STEP4[(i * 7) % 40] += ch
VARIANTS.append(('step4', '4-wave step (4 blocks + skeleton): ' + ' '.join(STEP4)) + block(STEP4, reps=1))

# what does ONE LDS operation cost inside a gap whose vector issue is (nearly) saturated?  base gaps: EEMC = 33 cycles, EMC = 25
for base in ['EEMC', 'EMC', 'EM']:
    for extra_ in ['R', 'W', 'L', 'S', 'RR', 'WW', 'LL', 'RW', 'RRRR', 'A', 'AA']:
        for pos in ('first', 'last'):
            slots = [(extra_ + base) if pos == 'first' else (base + extra_)] * 10
            VARIANTS.append((f'g_{base}_{extra_}_{pos}', f'every gap: {slots[0]}') + block(slots))
# LDS operations only in every other gap / every fifth gap
for base in ['EEMC', 'EMC']:
    for extra_ in ['R', 'W', 'L']:
        slots = [base + extra_, base] * 5
        VARIANTS.append((f'h_{base}_{extra_}', f'alternate gaps: {slots[0]} {slots[1]}') + block(slots))

# ---- round 5, second question: would the attention backward at TWO waves per SIMD (8 waves x 64 keys, no one-block lookahead:
# registers; the dQ product over ALL the workgroup's keys with 16x16x32 MFMAs so that no 32 x 32 fp32 partial per wave crosses LDS)
# be issue-bound where the model says?  One "step" of one wave = 2 blocks + the per-step part, as COARSE phases (what a stream
# without lookahead looks like: loads, products, the vector phase, products, stores) and as a fine interleave for comparison.
def sp8_step(fine):
    b = Body()
    def mf16():
        b.lines.append(f'v_mfma_f32_16x16x32_bf16 v[{ACC[0]}:{ACC[0] + 3}], v[{A}:{A + 3}], v[{B}:{B + 3}], v[{ACC[0]}:{ACC[0] + 3}]')
    n_mfma = 0
    for blk in range(2):
        if not fine:
            for _ in range(8): b.filler('dsr128')          # row constants straight into the score / dP registers
            if blk == 0:
                for _ in range(4): b.filler('dsr128')      # Q / dO rows of the step
                for _ in range(8): b.filler('dsr_tr')      # transposed Q / dO
            for _ in range(4): b.mfma(); n_mfma += 1
            for k in 'E' * 16 + 'M' * 16 + 'C' * 16: b.filler(KINDS[k])
            for _ in range(4): b.mfma(); n_mfma += 1
            for _ in range(4): b.filler('dsw')
        else:
            slots = [list(x) for x in BAL]
            extra = list('LLLLLLLL') + (list('LLLL' 'RRRRRRRR') if blk == 0 else [])
            for i, ch in enumerate(extra): slots[(2 * i) % 10].insert(0, ch)
            for sl in slots:
                b.mfma(); n_mfma += 1
                for ch in sl:
                    if ch in 'WR' and sl is not None and ch == 'R' and False: pass
                    b.filler(KINDS[ch])
    # the dQ phase of the previous step: 16 transposed reads, 8 MFMAs 16x16x32, the 2-way reduce and the atomics (as adds / nops)
    for i in range(8):
        b.filler('dsr_tr'); b.filler('dsr_tr')
        mf16()
    b.filler('dsr128')
    for _ in range(4): b.filler('add')
    for _ in range(4): b.filler('nop')
    return b, 20        # "gaps" = the 20 32x32x16 MFMAs of the two blocks: cycles per gap x 10 = cycles per block (the dQ MFMAs = 4 more gaps of pipe)


VARIANTS.append(('sp8_phase', 'two-waves-per-SIMD backward, one step, COARSE phases (per block: 10 gaps + 2 pipe-equivalents of dQ)') + sp8_step(False))
VARIANTS.append(('sp8_fine', 'the same instructions finely interleaved') + sp8_step(True))

clob = ', '.join(f'"v{i}"' for i in range(32, 180))

print('// GENERATED by tools/micro/gen_mfma_fillers.py -- do not edit.  See that file for what this measures.')
print('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <cstring>\n#include <vector>\n#include <algorithm>')
print('#define CLOB ' + clob + ', "memory"')
print('''
#define INIT() asm volatile("v_mov_b32 v96, 0x3c003c00\\n\\tv_mov_b32 v97, 0x3c003c00\\n\\tv_mov_b32 v98, 0x3c003c00\\n\\tv_mov_b32 v99, 0x3c003c00\\n\\t" \\
    "v_mov_b32 v100, 0x3c003c00\\n\\tv_mov_b32 v101, 0x3c003c00\\n\\tv_mov_b32 v102, 0x3c003c00\\n\\tv_mov_b32 v103, 0x3c003c00\\n\\t" \\
    "v_mov_b32 v104, 0xbf800000\\n\\tv_mov_b32 v105, 0x3f000000\\n\\tv_mov_b32 v106, 0x3f400000\\n\\tv_mov_b32 v107, 0x3f200000\\n\\t" \\
    "v_mov_b32 v108, 0x3f100000\\n\\tv_mov_b32 v109, 0x3f300000\\n\\tv_mov_b32 v110, 0x3f500000\\n\\tv_mov_b32 v111, 0x3f600000\\n\\t" \\
    "v_mov_b32 v178, 0\\n\\tv_mov_b32 v179, 0\\n\\t" \\
    "v_mov_b32 v176, %0\\n\\tv_mov_b32 v177, %1" :: "v"(a8), "v"(a16) : CLOB)
''')
for name, desc, body, gaps in VARIANTS:
    print(f'// {desc}')
    print(f'__global__ __launch_bounds__(512) void k_{name}(unsigned long long* out, int iters) {{')
    print('    extern __shared__ char lds[];')
    print('    const unsigned a8 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 512 * 0;')
    print('    const unsigned a16 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + 32768 + (threadIdx.x & 63) * 16;')
    print('    INIT();')
    print('    for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 v32, 0" ::: CLOB);')
    print('    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();')
    print('    const unsigned long long t0 = __builtin_amdgcn_s_memtime();')
    print('    for (int it = 0; it < iters; ++it) {')
    print(f'        asm volatile("{body.text()}\\n\\ts_waitcnt lgkmcnt(0)" ::: CLOB);')
    print('    }')
    print('    asm volatile("s_nop 15\\n\\ts_nop 15" ::: CLOB);')
    print('    const unsigned long long t1 = __builtin_amdgcn_s_memtime();')
    print('    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();')
    print('    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; }')
    print('}')

print('''
struct Var { const char* name; const char* desc; void (*fn)(unsigned long long*, int); int gaps; };
static const Var VARS[] = {''')
for name, desc, body, gaps in VARIANTS:
    print(f'    {{"{name}", "{desc}", k_{name}, {gaps}}},')
print('''};
int main(int argc, char** argv) {
    unsigned long long* d;
    (void)hipMalloc(&d, 1 << 20);
    std::vector<unsigned long long> h(2 * 256 * 8);
    const int iters = 4000;
    const char* only = argc > 1 ? argv[1] : nullptr;
    printf("%-14s %6s | %10s %10s %8s | %10s %10s %8s | %s\\n", "variant", "gaps", "cyc/gap 1w", "ns/gap 1w", "GHz", "cyc/gap 2w", "per SIMD", "GHz", "what");
    for (const Var& v : VARS) {
        if (only && !strstr(v.name, only)) continue;
        double res[2][3];
        for (int w2 = 0; w2 < 2; ++w2) {
            const int threads = w2 ? 512 : 256, nw = 256 * threads / 64;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(v.fn, dim3(256), dim3(threads), 100 * 1024, 0, d, iters);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h.data(), d, 2 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<double> cyc(nw), rt(nw);
            for (int i = 0; i < nw; ++i) { cyc[i] = (double)h[2 * i]; rt[i] = (double)h[2 * i + 1]; }
            std::sort(cyc.begin(), cyc.end());
            std::sort(rt.begin(), rt.end());
            const double c = cyc[nw / 2] / ((double)iters * v.gaps), ns = rt[nw / 2] * 10.0 / ((double)iters * v.gaps);
            res[w2][0] = c; res[w2][1] = ns; res[w2][2] = c / ns;
        }
        printf("%-14s %6d | %10.2f %10.2f %8.3f | %10.2f %10.2f %8.3f | %s\\n", v.name, v.gaps, res[0][0], res[0][1], res[0][2],
               res[1][0], res[1][0] / 2, res[1][2], v.desc);
        fflush(stdout);
    }
    return 0;
}''')
