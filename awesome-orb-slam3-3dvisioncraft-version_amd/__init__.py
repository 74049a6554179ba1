"""MI355X-native ORB-SLAM3 hot path (import as `orbhip`; see orbhip/__init__.py)."""
