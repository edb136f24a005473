"""Namespace package: `objectdetection.pytorch_amd` is the MI355X-native hot path."""
