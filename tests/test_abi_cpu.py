"""CPU-side checks of the drop-in boundary: the HIP library builds, loads, exports every
symbol include/quasar_slam.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT, load_pkg


def header_functions():
    txt = open(os.path.join(ROOT, "include", "quasar_slam.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    pkg = load_pkg()
    pkg.build()
    lib = pkg.load()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    from importlib import import_module
    sigs = import_module(pkg.__name__ + "._lib").SIGNATURES
    assert sorted(sigs) == names, "ctypes signatures and header disagree"


def test_config_struct_layout_and_defaults():
    pkg = load_pkg()
    lib = pkg.load()
    cfg = pkg.QsConfig()
    assert lib.qs_config_default(C.byref(cfg)) == 0
    # reference constants, dual_bot_mapper.py:57-99
    assert (cfg.size, cfg.res, cfg.ox, cfg.oy) == (200, 0.05, -5.0, -5.0)
    assert (cfg.min_dist, cfg.max_dist) == (0.05, 1.2)
    assert (cfg.closure_radius, cfg.min_poses_between, cfg.closure_correction) == (0.6, 30, 0.5)
    assert cfg.max_agent == 2 and cfg.enable_counts == 1
    assert lib.qs_version().startswith(b"quasar-slam-amd")


def test_protocol_sizes_and_packers():
    pkg = load_pkg()
    P = pkg.protocol
    assert (P.PACKET_SIZE, P.PACKET_SIZE_V1, P.ZONE_SIZE, P.TARGET_SIZE) == (42, 41, 20, 12)
    one = P.pack_packet(2, 1.5, -2.25, 0.5, 10, 20, 0.1, 0.2, 0.3, 0.4, 5)
    vec = P.pack_packets([2], [1.5], [-2.25], [0.5], [10], [20], [[0.1, 0.2, 0.3, 0.4]], [5])
    assert vec.shape == (1, 42) and vec.tobytes() == one
    assert struct.unpack(P.PACKET_FMT, one)[0] == b"QSRL"
    assert P.zone_packet(None) == struct.pack("<4sffff", b"ZONE", 999.0, 999.0, -999.0, -999.0)
    assert P.zone_packet((1, 2, 3, 4)).hex() == "5a4f4e450000803f000000400000404000008040"
    assert P.compute_bounding_box([], []) is None
    assert P.compute_bounding_box([1, -1], [2, 5]) == (-1, 2, 1, 5)
    buf, lens = P.pack_datagrams([one, one[:41], b"xy", one + b"\0" * 30])
    assert buf.shape == (4, 48) and lens.tolist() == [42, 41, 2, 72]


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = load_pkg()
    with pytest.raises(pkg.QuasarError, match="no HIP device"):
        pkg.QuasarMapper()


def test_product_package_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "distributed-multi-agent-slam-swarm-robotics-system_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and False, \
                    f"{f} mentions the oracle: the product path must not use it"


def test_legacy_v0_bridge_formats():
    """N4: the ROS bridge's 743-byte v0 packet and CMD1 reply (server_nodes/udp_bridge.py:25-38, :140-146)."""
    pkg = load_pkg()
    P = pkg.protocol
    assert P.PACKET_SIZE_V0 == 743 and struct.calcsize(P.CMD_FMT) == 12
    ranges = np.linspace(0.1, 4.0, 181, dtype=np.float32)
    raw = struct.pack(P.PACKET_FMT_V0, b"QSRL", 3, 1.0, -2.0, 0.5, 181, *ranges.tolist())
    agent, x, y, yaw, rg = P.unpack_v0(raw)
    assert (agent, x, y, yaw) == (3, 1.0, -2.0, 0.5) and (rg == ranges).all()
    assert P.unpack_v0(raw[:-1]) is None and P.unpack_v0(b"XXXX" + raw[4:]) is None
    assert P.pack_cmd(0.25, -1.0) == struct.pack("<4sff", b"CMD1", 0.25, -1.0)


def test_sparse_fuse_block_constants_match_the_cpu_stand_in():
    """tests/test_dist_cpu.py drives dist.sparse_fuse with a numpy adapter that mirrors the kernels' block / payload layout."""
    import test_dist_cpu as T
    txt = open(os.path.join(ROOT, "include", "quasar_slam.h")).read()
    w = int(re.search(r"#define QS_DIRTY_BLOCK_W (\d+)", txt).group(1))
    h = int(re.search(r"#define QS_DIRTY_BLOCK_H (\d+)", txt).group(1))
    assert (w, h) == (T.BW, T.BH) and w * h == 64          # one lane per cell of a block
