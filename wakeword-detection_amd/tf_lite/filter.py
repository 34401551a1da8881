from wwhip.filter import Filter  # noqa: F401
