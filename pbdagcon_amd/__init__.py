"""pbdagcon_amd -- MI355X-native DAGCon consensus (the pbdagcon hot path).

    capi        ctypes binding of include/dagcon.h (libdagcon_hip.so, HIP/gfx950)
    consensus   host-side mirror of the reference interface (Alignment,
                normalizeGaps, trimAln, AlnGraphBoost, CnsResult)
    synth       deterministic synthetic pileups
    m5          BLASR -m 5 reader / FASTA writer around the C ABI

The compute path is the HIP library only; importing this package does not
load it, calling into it without the built extension raises.
"""
__version__ = "0.1.0"
