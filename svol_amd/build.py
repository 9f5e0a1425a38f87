"""Build libsvol_hip.so (the C-ABI kernel library, include/svol_hip.h) in-tree with hipcc for gfx950.

    python -m svol_amd.build            # incremental
    python -m svol_amd.build --force

No CPU fallback exists: if this library is missing the product raises at first use.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libsvol_hip.so')
SOURCES = ['gemm.hip', 'gemm_bf16.hip', 'gemm_tn_bf16.hip', 'gemm_ws_bf16.hip', 'gemm_n256_bf16.hip', 'norm.hip', 'attention.hip', 'attention_bf16.hip', 'gate.hip', 'criterion.hip', 'posteval.hip', 'vit.hip', 'attn_weights.hip', 'resnet.hip', 'resnet_train.hip', 'blocks.hip', 'lsap_host.hip', 'heads.hip', 'clock_probe.hip']
# the MFMA-path files are written against h16_t (csrc/common.h) and compiled a second time with fp16 operands
H16_SOURCES = ['gemm_bf16.hip', 'gemm_tn_bf16.hip', 'gemm_ws_bf16.hip', 'gemm_n256_bf16.hip', 'attention_bf16.hip']
ARCH = 'gfx950'
COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + os.environ.get('SVOL_BUILD_DEFS', '').split()   # (SVOL_BUILD_DEFS: lab builds, e.g. -DSVOL_GELU_AS)
PER_FILE = {'criterion.hip': ['-ffp-contract=off'], 'posteval.hip': ['-ffp-contract=off'],  # bit-exact cost / LSAP arithmetic
            # SLP-packed f32 math (v_pk_mul_f32 on odd register pairs + v_mov/v_perm fix-ups) costs the VALU-bound
            # attention loops 20-30 % more vector instructions than the scalar form
            'attention_bf16.hip': ['-fno-slp-vectorize', '-Wno-inline-asm']}   # (-Wno-inline-asm: the LDS-DMA statements write M0, which hipcc reserves and warns about)


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hdrs = [os.path.join(CSRC, 'common.h'), os.path.join(os.path.dirname(HERE), 'include', 'svol_hip.h')]
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + COMMON + PER_FILE.get(src, []) + ['-c', s, '-o', o])
        if src in H16_SOURCES:
            o16 = os.path.join(objdir, src.replace('.hip', '_f16.o'))
            objs.append(o16)
            if force or _stale(o16, [s] + hdrs):
                jobs.append([hipcc] + COMMON + PER_FILE.get(src, []) + ['-DSVOL_H16_FP16', '-c', s, '-o', o16])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)
