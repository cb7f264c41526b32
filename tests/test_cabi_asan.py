"""CPU tier (sanitizer hook): the HOST side of the C ABI -- argument validation, (tile config, split-K, k-group)
selection incl. the tuned-table lookup, launch geometry, GroupNorm path selection -- built with AddressSanitizer
(``MD_ASAN=1 magicdance_amd/csrc/build.sh`` -> libmagicdance_hip_asan.so) and driven through every launcher in a subprocess that
preloads the sanitizer runtime.  No GPU: a launcher that gets past validation fails at its first HIP call and returns MD_ERR_HIP;
what is checked is that no host path reads or writes out of bounds on the way (ASAN aborts the process otherwise)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "magicdance_amd", "libmagicdance_hip_asan.so")

DRIVER = r'''
import ctypes as C, itertools, sys
sys.path.insert(0, %(root)r)
from magicdance_amd import _lib
lib = C.CDLL(%(lib)r)
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
buf = (C.c_char * (1 << 20))()
ptr = C.addressof(buf)
seen = set()
# md_igemm: every layer geometry class of the step x forced / automatic configs, k-groups and splits (host path: validate,
# choose incl. the tuned table, default_kg, fast_div magic, tile counts) -- the launch itself fails without a device
shapes = [(2, 64, 64, 320, 0, 320, 3, 1, 0), (3, 8, 8, 1280, 0, 1280, 3, 1, 0), (2, 16, 16, 1280, 1280, 1280, 3, 1, 0),
          (2, 32, 32, 640, 0, 640, 3, 2, 0), (2, 16, 16, 1280, 0, 1280, 3, 1, 1), (2, 1, 4096, 320, 0, 2560, 1, 1, 0),
          (16, 1, 256, 1280, 0, 1280, 1, 1, 0), (1, 64, 64, 8, 0, 320, 3, 1, 0), (1, 7, 9, 64, 0, 64, 3, 1, 0)]
for (b, h, w, c0, c1, n, ks, st, up), cfg, kg, sk in itertools.product(shapes, (-1, 4, 7, 12, 15, 24, 25, 27, 28, 33, 99),
                                                                       (0, 1, 2, 4, 3), (0, 1, 2, 8)):
    p = _lib.IgemmParams()
    p.a0, p.a1, p.c0, p.c1 = ptr, (ptr if c1 else 0), c0, c1
    p.batch, p.hin, p.win = b, h, w
    p.hout, p.wout = (2 * h, 2 * w) if up else (((h + 1) // 2, (w + 1) // 2) if st == 2 else (h, w))
    p.ksize, p.stride, p.ups, p.w, p.n = ks, st, up, ptr, n
    p.out, p.ld_out, p.n_tr_begin = ptr, n, n
    p.ws, p.ws_bytes = ptr, 1 << 20
    p.force_cfg, p.force_kg, p.force_splitk = cfg, kg, sk
    p.gn_part = ptr if (kg == 1 and sk == 0) else 0
    seen.add(lib.md_igemm(C.byref(p), None))
    lib.md_igemm_workspace_bytes(C.byref(p))
a = _lib.AttentionParams()
for d, nq, n0, n1, fp8 in itertools.product((40, 80, 160, 32, 64, 128, 8, 48), (4096, 77, 1), (4096, 77), (0, 4096), (0, 1)):
    a.q = a.k0 = a.vt0 = a.out = ptr
    a.k1 = a.vt1 = ptr if n1 else 0
    a.batch, a.heads, a.nq, a.d, a.n0, a.n1, a.n1_batches = 2, 8, nq, d, n0, n1, 1
    a.ld_q = a.ld_k0 = a.ld_k1 = a.ld_out = 8 * d
    a.ld_vt0, a.ld_vt1 = (n0 + 15) & ~15, (n1 + 15) & ~15
    a.q_batch_stride = a.k0_batch_stride = a.out_batch_stride = nq * 8 * d
    a.vt0_batch_stride = a.vt1_batch_stride = 8 * d * 4112
    a.kv_fp8, a.scale = fp8, 0.1
    seen.add(lib.md_attention(C.byref(a), None))
g = _lib.GroupNormParams()
for (bb, hw, c0, c1), parts in itertools.product(((2, 4096, 320, 0), (3, 4096, 640, 320), (2, 64, 1280, 1280), (1, 35, 512, 0), (1, 16, 8, 0)), (0, 1)):
    g.x0, g.x1, g.c0, g.c1 = ptr, (ptr if c1 else 0), c0, c1
    g.batch, g.hw, g.groups, g.eps = bb, hw, 32, 1e-5
    g.gamma = g.beta = g.out = g.ws = ptr
    g.ws_bytes = 1 << 20
    g.part0, g.part1 = (ptr, ptr if c1 else 0) if parts else (0, 0)
    seen.add(lib.md_groupnorm(C.byref(g), None))
    lib.md_groupnorm_wants_partials(bb, hw, c0 + c1, 32)
    lib.md_groupnorm_workspace_bytes(bb, hw, 32)
f = _lib.FfBlockParams()
for (m, c), head, dual, bm in itertools.product(((8192, 320), (12288, 320), (2048, 640), (65536, 320), (192, 1280), (100, 320)), (0, 1), (0, 1), (0, 32, 64, 128, 7)):
    f.x = f.x_lo = f.out = f.out_lo = f.w1 = f.s1 = f.s0 = f.w2 = f.b2 = ptr
    f.attn = f.wo = f.bo = ptr if head else 0
    f.w1_2 = f.s1_2 = f.s0_2 = f.w2_2 = f.b2_2 = ptr if dual else 0
    f.wo_2 = f.bo_2 = ptr if (dual and head) else 0
    f.m, f.c, f.ln_eps, f.m_split, f.force_bm = m, c, 1e-5, (m // 3 // 128 * 128 if dual else 0), bm
    seen.add(lib.md_ff_block(C.byref(f), None))
    lib.md_ff_block_supported(m, c)
seen.add(lib.md_layernorm(ptr, ptr, ptr, ptr, 77, 320, 1e-5, None))
seen.add(lib.md_softmax_rows(ptr, 4096, ptr, 4096, 8, 4096, 0.1, None))
seen.add(lib.md_add_f16(ptr, ptr, ptr, 4096, 1024, None))
seen.add(lib.md_gemv_f32(ptr, ptr, ptr, ptr, 50, 320, 1280, 1, None))
seen.add(lib.md_gather_rows(ptr, ptr, 4, 100, ptr, 0, 50, 5, 4000, ptr, None))
seen.add(lib.md_ddim_update(ptr, ptr, 4, ptr, None, ptr, ptr, ptr, None, 2, 4, 4096, None))
print("statuses", sorted(seen), lib.md_version())
'''


@pytest.mark.timeout(900)
def test_host_side_of_the_c_abi_is_clean_under_address_sanitizer(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("host-path driver for the build container: with a device present the launchers would run on its dummy buffers")
    rt = sorted(glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no clang AddressSanitizer runtime in this image")
    if not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(f)
                                     for f in glob.glob(os.path.join(ROOT, "magicdance_amd", "csrc", "*.h*")) +
                                     [os.path.join(ROOT, "include", "magicdance_hip.h")]):
        subprocess.check_call(["bash", os.path.join(ROOT, "magicdance_amd", "csrc", "build.sh")], env=dict(os.environ, MD_ASAN="1"))
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT, "lib": LIB}], env=env, capture_output=True, text=True,
                       timeout=800)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "statuses" in r.stdout and " 10" in r.stdout.splitlines()[-1]      # ABI version 10
    # validation outcomes only: bad argument / unsupported / workspace / no-device -- never MD_OK without a GPU
    st = eval(r.stdout.splitlines()[-1].split("statuses", 1)[1].rsplit("]", 1)[0] + "]")
    assert set(st) <= {-1, -2, -3, -4} and -4 in st, st
