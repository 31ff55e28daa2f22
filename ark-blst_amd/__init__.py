"""ark-blst_amd — MI355X-native BLS12-381 MSM backend (Python host binding over the C ABI).

The product is the C-ABI shared library `ark-blst_amd/lib/libarkblst_amd.so` (include/arkblst_amd.h); this
package is the thin ctypes host used by tests/ and bench.py, mirroring the reference's operator surface
for this path:  `<G1Projective as VariableBaseMSM>::msm(bases, scalars) -> Result<G1Projective, usize>`
(/root/reference/src/g1.rs:602-632, src/g2.rs:582-612).

There is NO CPU fallback: if the HIP library is missing or no GPU is present the calls raise.
The directory name contains a hyphen (it is the name the build contract asks for); import it with
`from __graft_entry__ import load_package; pkg = load_package()` or via importlib (see that helper).
"""
from .binding import (  # noqa: F401
    Context,
    MsmError,
    SCALAR_CANONICAL,
    SCALAR_MONTGOMERY,
    MAX_WINDOWS,
    final_exponentiation,
    fold_windows,
    test_plan,
    g1_sum,
    g2_sum,
    lib_path,
    load_library,
    profile_dict,
    RcclComm,
    rccl_unique_id,
    rccl_lib_path,
    load_rccl_library,
)
from .msm import G1Projective, G2Projective  # noqa: F401
