from wwhip.wakeword import WakewordTrigger  # noqa: F401
