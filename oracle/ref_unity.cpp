// oracle/ref_unity.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never linked into the product).
//
// Compile recipe for the *unmodified* reference ugs_sampler sources, read where they lie under
// /root/reference (nothing is copied into this repository).  The four reference translation units
// are pulled into one TU by #include so that a single g++ invocation (oracle/build_ref.py) builds
// the reference pybind11 module `ugs_sampler` into oracle/_ref/.
//
// The single deviation: the reference allocates its output tensors with
// `.pinned_memory(true)` (src/sampler.cpp:190, src/ugs_sampler_batch_extension.cpp:95), which
// raises "No HIP GPUs are available" in a GPU-less container.  The token is re-spelled to
// `.pinned_memory(false)` by a macro defined AFTER the torch headers have been parsed, so only
// those two call sites are affected.  This changes where the output bytes live, not a single
// arithmetic operation or any output value.
#include <torch/extension.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#define pinned_memory(x) pinned_memory(false)

#include "src/preproc.cpp"
#include "src/sampler.cpp"
#include "src/ugs_sampler_batch_extension.cpp"
#include "src/extension.cpp"
