from wwhip.keyword import KeywordRecognizer  # noqa: F401
