"""wwhip - MI355X-native wake-word inference hot path behind the reference's Python surface.

Importing this package touches no GPU and loads no native code; the HIP library is loaded on
first use and there is no CPU fallback for the hot path.
"""
__all__ = ["RingBuffer", "SpeechContext", "SpeechPipeline", "ActivationTimeout", "TFLiteModel", "Filter",
           "WakewordTrigger", "WakewordBank", "Engine", "StreamBank", "get_posterior", "far_frr",
           "ContextBank", "SpeechPipelineBank", "VadBank", "ActivationTimeoutBank"]


def __getattr__(name):
    import importlib
    table = {
        "RingBuffer": "ring_buffer", "SpeechContext": "context", "SpeechPipeline": "pipeline",
        "ActivationTimeout": "activation_timeout", "TFLiteModel": "models", "Filter": "filter",
        "WakewordTrigger": "wakeword", "WakewordBank": "wakeword", "Engine": "engine", "StreamBank": "engine",
        "get_posterior": "evaluate", "far_frr": "evaluate",
        "ContextBank": "context", "SpeechPipelineBank": "pipeline", "VadBank": "vad", "ActivationTimeoutBank": "activation_timeout",
    }
    if name in table:
        return getattr(importlib.import_module(f"{__name__}.{table[name]}"), name)
    raise AttributeError(name)
