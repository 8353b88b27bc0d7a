"""oracle: part of the MI355X-native DH-AUG hot path."""
