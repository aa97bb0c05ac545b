"""Model-side modules that call the sparse decode path (reference: quest/models/)."""
from .QuestAttention import QuestAttention  # noqa: F401
