from wwhip.io import WavInput  # noqa: F401
