from .AgentBase import AgentBase
from .AgentPPO import AgentPPO, ActorPPO, CriticPPO
