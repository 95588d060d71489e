"""Namespace package root; the product lives in ``efficientdet.pytorch_amd``."""
