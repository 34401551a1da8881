from wwhip.models import TFLiteModel  # noqa: F401
