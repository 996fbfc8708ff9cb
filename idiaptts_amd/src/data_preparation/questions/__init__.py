from .QuestionLabelGen import QuestionLabelGen  # noqa: F401
