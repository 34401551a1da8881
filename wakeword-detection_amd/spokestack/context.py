from wwhip.context import SpeechContext  # noqa: F401
