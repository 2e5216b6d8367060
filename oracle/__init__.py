"""CPU oracle for the Det3D hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (det3d_amd) never does: it fails loudly when the HIP
library is missing instead of falling back to anything in here.
"""
