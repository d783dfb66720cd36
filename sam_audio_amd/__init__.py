"""MI355X-native implementation of the SAMAudio.separate() hot path."""
