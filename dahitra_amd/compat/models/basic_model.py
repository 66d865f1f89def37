from dahitra_amd.models.basic_model import CDEvaluator  # noqa: F401
