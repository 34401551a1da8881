from wwhip.vad import (AGGRESSIVE, LOW_BITRATE, QUALITY, VERY_AGGRESSIVE, VoiceActivityDetector,  # noqa: F401
                       VoiceActivityTrigger)
