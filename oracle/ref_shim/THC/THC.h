/* Build shim, NOT reference code.  The reference's operators/src/segment_reduction.cpp starts with
 * `#include <THC/THC.h>`, a header modern torch no longer ships, and uses nothing from it (the file
 * only touches at::Tensor through <torch/extension.h>).  This empty header lets the UNMODIFIED
 * reference source compile where it lies (oracle/ref_build.py). */
