// oracle/ref_eps_unity.cpp -- TEST INFRASTRUCTURE ONLY.  Compile recipe for the unmodified reference
// epsilon_uniform_sampler source, read where it lies under /root/reference (nothing copied).  Same single deviation as
// oracle/ref_unity.cpp: `.pinned_memory(true)` (epsilon_uniform_sampler.cpp:181) is re-spelled `false` by a macro defined
// after the torch headers, because a GPU-less container cannot pin memory; no arithmetic is touched.
#include <torch/extension.h>
#include <pybind11/pybind11.h>

#define pinned_memory(x) pinned_memory(false)

#include "src/epsilon_uniform_sampler.cpp"
