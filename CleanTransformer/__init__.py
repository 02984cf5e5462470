"""Drop-in import paths of the reference package, backed by the MI355X-native implementation (cleantransformer_amd)."""
