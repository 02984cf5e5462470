from cleantransformer_amd.generation.generation_util import GenerationMixin  # noqa: F401
