"""The C-ABI library loads on a box without a GPU and exports every symbol include/avt.h declares; the ctypes
binding table covers the header one to one; host-only entry points behave.  No device calls here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch


def _header_functions(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(avt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(avt):
    names = _header_functions(avt._lib.HEADER_PATH)
    assert len(names) >= 14
    handle = ctypes.CDLL(avt._lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "libavt_hip.so does not export %s" % n
    assert sorted(list(avt._lib.SIGNATURES) + ["avt_last_error"]) == names
    assert avt._lib.lib().avt_abi_version() == avt._lib.ABI_VERSION == 8


def test_plane_job_table_layout_matches_the_header(avt):
    """AvtPlaneJob (include/avt.h) as the three parties see it: the header's fields mirrored in ctypes, the library's own sizeof, and the
    struct format train_ops._refresh_planes packs its device table with."""
    import struct

    class AvtPlaneJob(ctypes.Structure):
        _fields_ = ([(n, ctypes.c_void_p) for n in ("w", "hi", "lo", "wscale", "map")] +
                    [(n, ctypes.c_int32) for n in ("kind", "f16", "rows", "k", "cout", "taps", "cin", "nsel", "gx", "gy", "blk0", "pad_")] +
                    [("sel", ctypes.c_int32 * 32)])

    src = open(avt._lib.HEADER_PATH).read()
    body = src[src.index("typedef struct AvtPlaneJob {"):src.index("} AvtPlaneJob;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    declared = re.findall(r"\b([a-z_0-9]+)(?:\[32\])?\s*[;,]", body)
    assert declared == [f[0] for f in AvtPlaneJob._fields_], declared
    assert ctypes.sizeof(AvtPlaneJob) == avt._lib.lib().avt_weight_planes_job_bytes() == struct.calcsize("<5Q12i32i") == 216
    assert AvtPlaneJob.sel.offset == 88 and AvtPlaneJob.blk0.offset == 80


def test_no_torch_types_in_header(avt):
    src = open(avt._lib.HEADER_PATH).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # comments cite the torch calls each entry replaces
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code and 'extern "C"' in code


def test_product_never_imports_the_oracle(avt):
    root = os.path.dirname(avt._lib.LIB_PATH)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "avt_oracle" not in txt.replace(
                    "oracle/avt_oracle.c", ""), f


def test_cpu_tensors_are_rejected_not_emulated(avt):
    with pytest.raises(avt._lib.AvtError):
        avt.ops.l2norm_rows(torch.zeros(4, 8))
    with pytest.raises(avt._lib.AvtError):
        avt.ops.sim_gemm_nt(torch.zeros(4, 8), torch.zeros(4, 8), 0.1)
    with pytest.raises(avt._lib.AvtError):
        avt.texture.TextureEngine(torch.nn.Linear(2, 2), torch.nn.Linear(2, 2), window=4, stride=2, device="cpu")


@pytest.mark.parametrize("W", [2, 5, 10, 13, 15, 20, 32, 33, 64, 100])
def test_sample_table_equals_torch_linspace(avt, W):
    fast, slow = avt.ops.clip_sample_table(W)
    ref_fast = torch.linspace(0, W - 1, 32).long()
    ref_slow = ref_fast[torch.linspace(0, 31, 8).long()]
    assert np.array_equal(fast, ref_fast.numpy()) and np.array_equal(slow, ref_slow.numpy())


def test_clip_pack_plan_is_the_inverse_of_the_sample_table(avt):
    W, S, n = 20, 4, 9
    starts = np.arange(n) * S
    F_ = starts[-1] + W + 2
    off, slot = avt.ops.clip_pack_plan(starts, W, F_)
    fast, slow = avt.ops.clip_sample_table(W)
    assert off[0] == 0 and off[-1] == n * 40 and len(slot) == n * 40
    seen = set()
    for f in range(F_):
        for v in slot[off[f] : off[f + 1]]:
            win, s = divmod(int(v), 40)
            src = starts[win] + (slow[s] if s < 8 else fast[s - 8])
            assert src == f
            seen.add(int(v))
    assert seen == set(range(n * 40))
    with pytest.raises(avt._lib.AvtError):
        avt.ops.clip_pack_plan(np.array([F_ - 3]), W, F_)  # window leaves the video: loud error


def test_conv_ktab(avt):
    tab = avt.ops.conv3d_ktab(16, (1, 3, 3), 10, 12, 16)
    assert tab.shape == (8 * 3 + 2, 2)  # K = 144 -> 3 K-steps of 64, + 16 zero bytes
    kc = 5  # chunk 5: tap 2 (dh=0, dw=2), channels 8..15
    assert tab[kc, 1] == (1 | 1 << 8 | 1 << 18) and tab[kc, 0] == 2 * 16 + 8
    assert (tab[18:24, 1] == -1).all() and (tab[24:] == 0).all()
