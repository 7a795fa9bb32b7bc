"""tvretrieval_amd: MI355X-native XML corpus-level moment-retrieval hot path."""
__version__ = "0.1.0"
