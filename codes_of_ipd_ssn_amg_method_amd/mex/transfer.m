function [Ac,Pro,As,indC] = transfer(varargin)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[Ac,Pro,As,indC] = ipd_mex('transfer', varargin{:});
end
