"""rebel_amd -- MI355X-native self-play data generation for ReBeL on Liar's Dice.

    rebel_amd.capi    ctypes binding of the C ABI (include/rebel_hip.h, rebel_amd/librebel_hip.so)
    rebel_amd.rela    drop-in for the reference's pybind11 module `cfvpy.rela`
                      (/root/reference/csrc/liars_dice/rela/pybind.cc:119-213), built by `make -C rebel_amd/csrc rela`

There is no CPU fallback: every compute entry point needs librebel_hip.so and a gfx950 device and raises otherwise.
"""
__version__ = "0.1.0"
