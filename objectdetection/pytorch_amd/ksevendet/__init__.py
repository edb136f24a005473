"""Drop-in for the reference's `ksevendet` package (the detector `train_ksevendet.py:207` builds): the default
configuration family — ResNet backbone (`variant: resnet18/34/50/101/152`), FPN / PANet-FPN neck, simple shared
heads with per-level BatchNorm — on the MI355X launch plans.  SURVEY.md 8 f-3."""
