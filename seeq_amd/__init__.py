"""seeq_amd -- MI355X-native per-line approximate DNA/RNA pattern matching.

A from-scratch HIP/gfx950 implementation of the hot path of ezorita/seeq
behind that project's own interfaces:

  * C:      include/libseeq.h, include/seeq.h  (seeq_amd/lib/libseeq_amd.so, seeq_amd/bin/seeq)
  * Python: this package exposes the reference module's surface --
            compile(), SeeqObject, SeeqMatch, SeeqIter, __version__
  * batch:  seeq_amd.device (include/seeq_amd.h) for device-resident buffers

All matching runs in HIP kernels; there is no CPU fallback.
"""
from . import _capi
from .module import (SeeqIter, SeeqMatch, SeeqObject, compile, exception, libseeq_exception,  # noqa: F401,A004
                     __version__)

build = _capi.build
