"""Entry points on top of the codecs (reference entrypoints/)."""
