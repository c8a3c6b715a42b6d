"""omg_amd — MI355X-native hot path of OMG (SDXL UNet forward, prompt-to-prompt attention
replacement, region-masked noise fusion).  See DESIGN.md."""
__version__ = "0.1.0"
