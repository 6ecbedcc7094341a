"""elementary_amd — MI355X-native block-render engine behind Elementary's Runtime API."""
