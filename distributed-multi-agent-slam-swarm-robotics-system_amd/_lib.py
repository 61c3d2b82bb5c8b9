"""ctypes binding of csrc/libquasar_slam.so (C ABI: include/quasar_slam.h).

There is no CPU fallback: if the HIP library is missing or no GPU is present the product
path raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("QUASAR_SLAM_LIB") or os.path.join(CSRC, "libquasar_slam.so")   # env: A/B builds of the same ABI
HEADER = os.path.join(os.path.dirname(_HERE), "include", "quasar_slam.h")

QS_CNT_NAMES = ("datagrams", "accepted", "rays", "cells", "hits", "closures", "landmarks", "rebases",
                "slam_windows", "slam_rounds", "slam_node_iters", "slam_misc_iters", "slam_cycles",
                "slam_realtime_100mhz", "slam_cyc_prepare", "slam_cyc_query", "slam_cyc_commit", "ekf_wrap_clamp", "edge_rays", "edge_overflow")
QS_STAGE_NAMES = ("decode", "slam", "raycast", "ekf", "slam_chain", "rc_rays", "rc_sort", "rc_raster")
UINT64_MAX = (1 << 64) - 1


class QsConfig(C.Structure):
    """struct qs_config (include/quasar_slam.h)."""
    _fields_ = [
        ("size", C.c_int32),
        ("res", C.c_double), ("ox", C.c_double), ("oy", C.c_double),
        ("separation", C.c_double),
        ("min_dist", C.c_double), ("max_dist", C.c_double),
        ("closure_radius", C.c_double),
        ("min_poses_between", C.c_int32),
        ("closure_correction", C.c_double),
        ("max_agent", C.c_int32),
        ("bots_per_graph", C.c_int32),
        ("enable_counts", C.c_int32),
        ("enable_ekf", C.c_int32),
        ("ekf_metres_per_tick", C.c_double),
        ("device", C.c_int32),
        ("raycast_mode", C.c_int32),
        ("seq_stride", C.c_int32),
        ("shard_bots", C.c_int32),
        ("shard_rank", C.c_int32),
        ("exact_trig", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class QuasarError(RuntimeError):
    pass


def build(force=False, quiet=True):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise QuasarError("building libquasar_slam.so failed:\n" + res.stdout[-4000:])
    if not quiet:
        print(res.stdout)
    return LIB_PATH


_lib = None

_vp, _i32, _i64, _u64, _f64, _f32, _sz = (C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double,
                                          C.c_float, C.c_size_t)

# name -> (restype, argtypes); every function include/quasar_slam.h declares
SIGNATURES = {
    "qs_version": (C.c_char_p, []),
    "qs_config_default": (_i32, [C.POINTER(QsConfig)]),
    "qs_create": (_i32, [C.POINTER(QsConfig), C.POINTER(_vp)]),
    "qs_destroy": (_i32, [_vp]),
    "qs_last_error": (C.c_char_p, [_vp]),
    "qs_set_stream": (_i32, [_vp, _vp]),
    "qs_sync": (_i32, [_vp]),
    "qs_reset": (_i32, [_vp]),
    "qs_set_bot_offset": (_i32, [_vp, _i32, _f64]),
    "qs_ingest": (_i32, [_vp, _vp, _sz, _sz, _vp, _vp, _u64]),
    "qs_ingest_device": (_i32, [_vp, _vp, _sz, _sz, _vp, _vp, _u64]),
    "qs_last_batch": (_i32, [_vp, _vp, _vp, _sz]),
    "qs_last_hits": (_i32, [_vp, _vp, _vp, _sz]),
    "qs_update_rays": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _u64]),
    "qs_world_to_grid": (_i32, [_vp, _vp, _sz, _i32, _vp]),
    "qs_grid_i8": (_i32, [_vp, _vp]),
    "qs_grid_i8_device": (_i32, [_vp, _vp]),
    "qs_grid_counts": (_i32, [_vp, _vp, _vp]),
    "qs_grid_logodds": (_i32, [_vp, _f32, _f32, _f32, _f32, _vp]),
    "qs_device_buffers": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz)]),
    "qs_slam_sizes": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "qs_slam_closures": (_i32, [_vp, _i32, _vp, _vp, _sz]),
    "qs_slam_closure_agents": (_i32, [_vp, _i32, _vp, _sz]),
    "qs_slam_landmarks": (_i32, [_vp, _i32, _vp, _vp, _sz]),
    "qs_slam_add_poses": (_i32, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "qs_drift": (_i32, [_vp, _i32, _vp]),
    "qs_zone": (_i32, [_vp, _i32, _vp, C.POINTER(_i32)]),
    "qs_zone_packet": (_i32, [_vp, _i32, _i32, _vp]),
    "qs_fuse": (_i32, [_vp, C.POINTER(_vp), _sz]),
    "qs_fuse_buffers": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), _sz]),
    "qs_fuse_buffers_range": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_vp), _sz, _sz, _sz, _i32]),
    "qs_fused_counts": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "qs_counts_source": (_i32, [_vp, _i32]),
    "qs_epoch_query": (_i32, [_vp, _u64, _sz, C.POINTER(_i32)]),
    "qs_mark_fused": (_i32, [_vp]),
    "qs_dirty_tracking": (_i32, [_vp, _i32]),
    "qs_dirty_blocks": (_i32, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "qs_sparse_fuse_begin": (_i32, [_vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_sz)]),
    "qs_sparse_fuse_plan": (_i32, [_vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "qs_sparse_fuse_apply": (_i32, [_vp]),
    "qs_fused_counts_buffer": (_i32, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "qs_rccl_unique_id": (_i32, [_vp]),
    "qs_rccl_comm_init": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "qs_rccl_comm_destroy": (_i32, [_vp]),
    "qs_sparse_fuse_rccl": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "qs_grid_to_pcd": (_i32, [_vp, _vp, _i32, _i32, _f64, _f64, _f64, _vp, _sz, C.POINTER(_sz)]),
    "qs_rasterise": (_i32, [_vp, _vp, _sz, _f64, _vp, _vp, _vp]),
    "qs_icp": (_i32, [_vp, _vp, _sz, _vp, _sz, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp]),
    "qs_nn_search": (_i32, [_vp, _vp, _sz, _vp, _sz, _f64, _i32, _vp, _vp, _vp]),
    "qs_diag_mfma_f64_rate": (_i32, [_vp, C.POINTER(_f64)]),
    "qs_diag_latencies": (_i32, [_vp, _vp]),
    "qs_set_chain_form": (_i32, [_vp, _i32]),
    "qs_chain_form": (_i32, [_vp]),
    "qs_voxel_downsample": (_i32, [_vp, _vp, _sz, _f64, _vp, _sz, C.POINTER(_sz)]),
    "qs_frontier_cells": (_i32, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "qs_frontier_members": (_i32, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "qs_frontier_clusters": (_i32, [_vp, _i32, _vp, _sz, C.POINTER(_sz)]),
    "qs_ekf_init": (_i32, [_vp, _i32, _f64, _vp]),
    "qs_ekf_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32]),
    "qs_ekf_state": (_i32, [_vp, _i32, _vp, _vp]),
    "qs_counters": (_i32, [_vp, _vp]),
    "qs_timing_enable": (_i32, [_vp, _i32]),
    "qs_stage_times": (_i32, [_vp, _vp, _vp, _i32]),
}


def load():
    """dlopen the HIP library and declare every entry point.  Raises if it is absent.
    A process that also uses torch must import torch BEFORE the first call of this function: the torch wheel bundles its
    own HIP runtime, and once the system's (which this library links) has initialised the GPU, torch's reports
    "No HIP GPUs are available".  dist.py and bench.py import torch first; a host that only maps never needs it."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QuasarError(
            f"{LIB_PATH} not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the mapper path.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)            # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(ctx, rc, what):
    if rc != 0:
        msg = load().qs_last_error(ctx)
        raise QuasarError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
