from wwhip.ring_buffer import RingBuffer  # noqa: F401
