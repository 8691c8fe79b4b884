from .fastspeech2 import FeedForwardTransformer  # noqa: F401
