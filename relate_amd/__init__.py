"""relate_amd: MI355X-native chromosome painting (the Paint stage of Relate).

The compute path is librelate_gpu.so (hand-written gfx950 kernels behind the C
ABI of include/relate_gpu.h); this package is the thin host side: ctypes
bindings, a mirror of the reference's Data / FastPainting / Paint interface,
synthetic-chunk helpers and the multi-GPU sharding over torch.distributed.
There is no CPU implementation here: without the built library, or without a
GPU, every compute call raises."""
from ._lib import RelateError, lib, library_path, record_bytes  # noqa: F401
from .paint import Data, FastPainting, Plan, paint_chunk, paint_stones  # noqa: F401
from .synth import synth_chunk, window_boundaries, write_chunk  # noqa: F401
from .topology import DistanceMeasure, NativeDistanceMeasure  # noqa: F401
from .treebuilder import Candidate, MinMatchScan, apply_candidates, minmatch_threshold  # noqa: F401
