"""MI355X-native hot path of the MPM garment simulator.

Importing the package has no side effects on the process (round 5: it used to set GPU_MAX_HW_QUEUES for every HIP user of the host
process).  The one code path that wants one hardware queue per HIP stream -- the four concurrent solver contexts of the
finite-difference training step -- asks for it itself: ``mpmavatar_amd.fd.request_hw_queues()``, INTEGRATION.md "hardware queues".
"""
