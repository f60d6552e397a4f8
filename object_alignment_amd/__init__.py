"""MI355X-native ICP alignment engine (hot path of patmo141/object_alignment)."""
