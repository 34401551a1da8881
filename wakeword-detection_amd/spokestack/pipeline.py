from wwhip.pipeline import SpeechPipeline  # noqa: F401
