"""MI355X-native (gfx950) FastSpeech2 acoustic-model forward path.

Drop-in for the model API of francissfy/fastspeech2: `network.fastspeech2.FeedForwardTransformer`,
`param.load_param.load_config` + `param/fastspeech.yaml`, `run.fastspeech_decode.decode`.
Behind it: hand-written HIP kernels in `csrc/` reached through the C ABI of `include/fs2_hip.h`.
"""
__version__ = "0.1.0"
