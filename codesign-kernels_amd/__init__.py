"""codesign-kernels_amd -- MI355X-native E3SM-MMF 2D MPDATA tracer advection.

Host-side mirror (Python) of the one routine this project replaces:
`advect_scalar2D(f,u,w,rho,rhow,flux)` of the reference
mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90 (:72, :247, :477),
bound to libmpdata_hip.so (hand-written HIP for gfx950) through the C-ABI in
include/mpdata_hip.h.  The Fortran driver + shim that north_star asks for live
in fortran/; this Python face exists for bench.py, the tests and
torch.distributed plumbing.  There is no CPU fallback here: if the HIP library
is missing or no device is usable, calls raise.
"""
from .capi import (EINVAL, EUNSUPPORTED, ESTATE, ECOMM, LAYOUT_REFERENCE, LAYOUT_WAVEMAJOR, MpdataError, Plan, VARIANT_EXACT, VARIANT_FAST, advect_scalar2D,
                   advect_scalar2D_host, release_host_buffers, algorithmic_bytes, build_library, debug_stages, device_count,
                   empty_staggered, fill_synthetic, get_variant, host_shapes, lib, lib_path, pack_shard,
                   set_plan_layout, set_serpentine, set_tile, set_variant, set_wm_flags, version, WMF_NOSTREAM, WMF_TPW1, WMF_NOSPLIT, WMF_SPLIT, shard_range, shapes, stage_shapes, stream_ceiling, unpack_shard)
from .shard import gather_outputs, partition, scatter_inputs

__all__ = ["EINVAL", "EUNSUPPORTED", "ESTATE", "ECOMM", "set_serpentine", "set_wm_flags", "version", "WMF_NOSTREAM", "WMF_TPW1", "WMF_NOSPLIT", "WMF_SPLIT", "LAYOUT_REFERENCE", "LAYOUT_WAVEMAJOR", "host_shapes", "set_plan_layout", "shard_range", "stream_ceiling", "MpdataError", "Plan", "VARIANT_EXACT", "VARIANT_FAST", "advect_scalar2D",
           "advect_scalar2D_host", "release_host_buffers", "algorithmic_bytes", "build_library", "debug_stages", "device_count",
           "empty_staggered", "fill_synthetic", "get_variant", "lib", "lib_path", "pack_shard", "set_tile",
           "set_variant", "shapes", "stage_shapes", "unpack_shard", "partition", "scatter_inputs",
           "gather_outputs"]
