"""quest_amd -- MI355X-native implementation of Quest's query-aware sparse decode path.

``quest_amd._kernels`` is the drop-in for the reference's PyBind module ``quest._kernels``;
``quest_amd.utils`` mirrors ``quest.utils``; ``quest_amd.models.QuestAttention`` mirrors
``quest.models.QuestAttention``.  Importing ``quest_amd.utils`` loads libquest_hip.so and fails
loudly when it has not been built -- there is no CPU fallback.
"""
__version__ = "0.1.0"
