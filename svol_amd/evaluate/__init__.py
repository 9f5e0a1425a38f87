"""Mirror of the reference's ``lib.evaluate`` package on the MI355X kernels."""
