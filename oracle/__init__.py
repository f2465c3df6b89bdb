"""CPU checker for the HIP sweeps -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this
package.  Nothing under fdtd_amd/ does, and the product never falls back to it.

`cpu`  ctypes front-end of oracle/liboracle.so (this repo's restatement, fdtd_oracle.c)
`ref`  ctypes front-end of oracle/_ref/libfdtd_ref_strict.so (the reference's own golden
       loops compiled by build_ref.sh); present only after that build has run.
"""
