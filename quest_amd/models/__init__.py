"""Model-side modules that call the sparse decode path (reference: quest/models/)."""
