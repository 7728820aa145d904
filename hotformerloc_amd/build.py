"""Build libhotformerloc_hip.so (gfx950) in-tree with hipcc.

`python -m hotformerloc_amd.build` or `__graft_entry__.build()`.  hipcc
cross-compiles without a GPU; the resulting .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIBNAME = 'libhotformerloc_hip.so'
ARCH = 'gfx950'
SOURCES = ['capi.hip', 'dwconv.hip', 'octree.hip', 'preprocess.hip', 'window_misc.hip', 'attention.hip', 'gemm_x3.hip', 'gemm_x6.hip', 'mlp_fused.hip', 'qkv_fused.hip', 'attn_fused.hip', 'attn_ws.hip', 'attn_pool.hip', 'wgrad_x3.hip', 'tapconv.hip', 'gemm_lt.hip', 'loss.hip']
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc',
         '-Wall', '-Wno-unused-function']
# hipBLASLt for hfl_gemm_bf16 (the ROCm copy that matches the headers; rpath so the loader finds it)
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')
LINK_FLAGS = ['-L' + os.path.join(ROCM, 'lib'), '-lhipblaslt', '-Wl,-rpath,' + os.path.join(ROCM, 'lib')]
# per-file extras.  attention.hip: MFMA results feed VALU softmax code directly, so keep the MFMA
# destination in arch VGPRs (the default heuristic parks it in AGPRs and pays a v_accvgpr_read per score)
EXTRA_FLAGS = {'attention.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1'],
               # mlp_fused.hip: element-wise math runs BETWEEN MFMAs there; packed f32 VALU (what SLP vectorisation of adjacent
               # scalar ops produces) stalls the matrix pipe, plain VALU does not
               'mlp_fused.hip': ['-fno-slp-vectorize'],
               'qkv_fused.hip': ['-fno-slp-vectorize'],
               'attn_fused.hip': ['-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1'],
               'attn_ws.hip': ['-fno-slp-vectorize', '-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm >= 7.0 for gfx950)')


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, 'hfl_common.h'), os.path.join(CSRC, 'x3_math.h'), os.path.join(CSRC, 'stage_stream.h'),
               os.path.join(HERE, '..', 'include', 'hotformerloc_hip.h')]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        if force or not _newer(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + os.environ.get('HFL_EXTRA_HIPCC_FLAGS', '').split() + \
            ['-c', s, '-o', o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (s, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in SOURCES]
    for stale in set(os.listdir(objdir)) - {os.path.basename(o) for o in objs}:      # objects whose source is gone
        os.remove(os.path.join(objdir, stale))
    lib = os.path.join(LIBDIR, LIBNAME)
    if force or jobs or not os.path.exists(lib):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', lib] + objs + LINK_FLAGS
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr)
    if verbose:
        print('built', lib)
    return lib


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
