from wwhip.activation_timeout import ActivationTimeout  # noqa: F401
